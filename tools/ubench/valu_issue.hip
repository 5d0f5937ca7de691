// Microbenchmark: what a vector instruction costs in SIMD issue cycles on
// gfx950 at 1, 2 and 4 waves per SIMD, for the instruction kinds the MOG fast
// loop is made of.  Prices the VALU-issue roofline of mog_batch2_k
// (bench.py).  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/valu_issue tools/ubench/valu_issue.hip
//   tools/ubench/valu_issue            (on the GPU box)
//
// Each wave runs ITER x 64 independent instructions of one kind and stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop.
// Residency is fixed by construction: one workgroup of 256 x w threads per CU
// (96 KB of LDS each, so no two share a CU; 256 workgroups) puts w waves on
// every SIMD; 8 waves per SIMD = two 1024-thread workgroups per CU.
// cycles per instruction per SIMD = (loop cycles of a wave) / (instructions
// per wave x waves per SIMD).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                              \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

constexpr int ITER = 2000;

// 16 instructions on 8 rotating destination registers: no instruction
// depends on the one before it
#define REP16(a)                                                              \
    a(0) a(1) a(2) a(3) a(4) a(5) a(6) a(7) a(0) a(1) a(2) a(3) a(4) a(5)     \
        a(6) a(7)
#define REP64(a) REP16(a) REP16(a) REP16(a) REP16(a)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void issue_k(unsigned long long *stamps,
                                                float *sink, float seed) {
    extern __shared__ float lds_hold[];
    f2 r[8];
    float s[8];
    for (int i = 0; i < 8; ++i) {
        r[i] = f2{seed + i, seed - i};
        s[i] = seed * (i + 1);
    }
    const f2 x = {seed * 0.5f, seed * 0.25f};
    const float y = seed * 0.125f;
    unsigned u = __float_as_uint(seed) + threadIdx.x;
    unsigned long long t0, t1, r0, r1;
    if (seed == 77.f) lds_hold[threadIdx.x] = seed;      // keep the LDS claim
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {            // v_fma_f32
#define A(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(y), "v"(x.x));
            REP64(A)
#undef A
        } else if (KIND == 1) {     // v_pk_fma_f32
#define A(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(x));
            REP64(A)
#undef A
        } else if (KIND == 2) {     // v_pk_mul_f32
#define A(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(x));
            REP64(A)
#undef A
        } else if (KIND == 3) {     // v_pk_add_f32
#define A(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(x));
            REP64(A)
#undef A
        } else if (KIND == 4) {     // v_add_f32
#define A(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(y));
            REP64(A)
#undef A
        } else if (KIND == 5) {     // v_max_f32
#define A(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(y));
            REP64(A)
#undef A
        } else if (KIND == 6) {     // v_cvt_f32_ubyte1
#define A(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(s[i]) : "v"(u));
            REP64(A)
#undef A
        } else if (KIND == 7) {     // v_rcp_f32
#define A(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
            REP64(A)
#undef A
        } else if (KIND == 8) {     // v_mov_b64
#define A(i) asm volatile("v_mov_b64 %0, %1" : "=v"(r[i]) : "v"(x));
            REP64(A)
#undef A
        } else if (KIND == 9) {     // v_cmp_lt_f32 + v_cndmask_b32 pairs
#define A(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(y) : "vcc");
            REP16(A) REP16(A)
#undef A
        } else if (KIND == 10) {    // the MOG fast step's mix: 54 pk, 40 plain, 2 rcp (per 96)
#define P(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(x), "v"(x));
#define Q(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(y));
#define R(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
            // 32 instructions: 18 pk, 13 plain, 1 rcp (interleaved)
            P(0) Q(0) P(1) P(2) Q(1) P(3) Q(2) P(4) P(5) Q(3) P(6) Q(4) P(7)
            P(0) Q(5) P(1) Q(6) P(2) P(3) Q(7) P(4) Q(0) P(5) P(6) Q(1) P(7)
            Q(2) P(0) Q(3) P(1) Q(4) R(5)
            P(0) Q(0) P(1) P(2) Q(1) P(3) Q(2) P(4) P(5) Q(3) P(6) Q(4) P(7)
            P(0) Q(5) P(1) Q(6) P(2) P(3) Q(7) P(4) Q(0) P(5) P(6) Q(1) P(7)
            Q(2) P(0) Q(3) P(1) Q(4) R(5)
#undef P
#undef Q
#undef R
        } else if (KIND == 12) {    // v_mul_f32
#define A(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(y));
            REP64(A)
#undef A
        } else if (KIND == 13) {    // v_cndmask_b32 alone (vcc set once)
#define A(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(y) : "vcc");
            REP64(A)
#undef A
        } else if (KIND == 14) {    // v_cmp_lt_f32 alone
#define A(i) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : : "v"(s[i]), "v"(y) : "vcc");
            REP64(A)
#undef A
        } else if (KIND == 15) {    // alternating v_pk_fma_f32 / v_add_f32
#define A(i) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4" : "+v"(r[i]), "+v"(s[i]) : "v"(x), "v"(x), "v"(y));
            REP16(A) REP16(A)
#undef A
        } else if (KIND == 16) {    // alternating v_pk_fma_f32 / v_max_f32
#define A(i) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_max_f32 %1, %1, %4" : "+v"(r[i]), "+v"(s[i]) : "v"(x), "v"(x), "v"(y));
            REP16(A) REP16(A)
#undef A
        } else if (KIND == 17) {    // v_fma_f32 x2 on the halves of a pair (= one v_pk_fma_f32 unpacked)
#define A(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(s[i]), "+v"(r[i].x) : "v"(y), "v"(x.x));
            REP16(A) REP16(A)
#undef A
        } else if (KIND == 18) {    // s_add_u32 on rotating registers
            unsigned q0 = 1, q1 = 2, q2 = 3, q3 = 4;
#define A(i) asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3) : : "scc");
            REP16(A)
#undef A
            if (q0 + q1 + q2 + q3 == 0x7fffffffu) s[0] += 1.f;
        } else if (KIND == 19) {    // s_nop 0
#define A(i) asm volatile("s_nop 0");
            REP64(A)
#undef A
        } else if (KIND == 20) {    // v_pk_fma_f32 + s_add_u32 (pair)
            unsigned q0 = 1;
#define A(i) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\ts_add_u32 %1, %1, 1" : "+v"(r[i]), "+s"(q0) : "v"(x), "v"(x) : "scc");
            REP16(A) REP16(A)
#undef A
            if (q0 == 0x7fffffffu) s[0] += 1.f;
        } else if (KIND == 21) {    // v_pk_fma_f32 + 2 x s_add_u32 (triple)
            unsigned q0 = 1, q1 = 2;
#define A(i) asm volatile("v_pk_fma_f32 %0, %3, %4, %0\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1" : "+v"(r[i]), "+s"(q0), "+s"(q1) : "v"(x), "v"(x) : "scc");
            REP16(A) REP16(A)
#undef A
            if (q0 + q1 == 0x7fffffffu) s[0] += 1.f;
        } else if (KIND == 22) {    // v_add_f32 + s_add_u32 (pair)
            unsigned q0 = 1;
#define A(i) asm volatile("v_add_f32 %0, %0, %2\n\ts_add_u32 %1, %1, 1" : "+v"(s[i]), "+s"(q0) : "v"(y) : "scc");
            REP16(A) REP16(A)
#undef A
            if (q0 == 0x7fffffffu) s[0] += 1.f;
        } else if (KIND == 23) {    // s_cmp + s_cbranch not taken
            const unsigned us = __builtin_amdgcn_readfirstlane(u);
#define A(i) asm volatile("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 1f\n\t1:" : : "s"(us) : "scc");
            REP16(A) REP16(A)
#undef A
        } else if (KIND == 11) {    // v_pk_fma_f32 with an SGPR-pair operand
#define A(i) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(r[i]) : "s"(x));
            REP64(A)
#undef A
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y + s[i];
    if (acc == 12345.678f) sink[0] = acc;       // keep the results alive
    if ((threadIdx.x & 63) == 0) {
        const unsigned wave = blockIdx.x * 16 + (threadIdx.x >> 6);
        stamps[2 * wave] = t1 - t0;
        stamps[2 * wave + 1] = r1 - r0;
    }
}

template <int KIND>
static void run(const char *name, int per_iter, unsigned long long *d_stamps,
                float *d_sink) {
    printf("%-36s", name);
    for (int w : {1, 2, 4, 8}) {
        const int threads = w == 8 ? 1024 : 256 * w;
        const int blocks = w == 8 ? 512 : 256;
        const size_t lds = w == 8 ? 72 * 1024 : 96 * 1024;
        const int waves_per_block = threads / 64;
        CHECK(hipFuncSetAttribute((const void *)issue_k<KIND>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
        std::vector<unsigned long long> h(2 * 16 * 512);
        double best = 1e30, mhz = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(d_stamps, 0, sizeof(h[0]) * h.size()));
            hipLaunchKernelGGL(issue_k<KIND>, dim3(blocks), dim3(threads), lds,
                               0, d_stamps, d_sink, 1.0009765f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), d_stamps, sizeof(h[0]) * h.size(),
                            hipMemcpyDeviceToHost));
            std::vector<double> d;
            double cyc = 0, real = 0;
            for (int b = 0; b < blocks; ++b)
                for (int v = 0; v < waves_per_block; ++v) {
                    d.push_back((double)h[2 * (b * 16 + v)]);
                    cyc += (double)h[2 * (b * 16 + v)];
                    real += (double)h[2 * (b * 16 + v) + 1];
                }
            std::sort(d.begin(), d.end());
            const double med = d[d.size() / 2];
            if (med < best) {
                best = med;
                mhz = cyc / real * 100.0;
            }
        }
        const double n = (double)ITER * per_iter;
        // cycles per instruction per SIMD, and the same in nanoseconds at the
        // clock the chip held in that run
        printf(" | w%d %5.2f cyc %4.0f MHz %5.2f ns", w, best / (n * w), mhz,
               best / (n * w) / mhz * 1e3);
    }
    printf("\n");
}

int main() {
    unsigned long long *d_stamps;
    float *d_sink;
    CHECK(hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 2 * 16 * 512));
    CHECK(hipMalloc((void **)&d_sink, 64));
    printf("shader cycles (s_memtime) per instruction per SIMD; wN = N waves "
           "per SIMD, all SIMDs of the chip busy\n");
    run<0>("v_fma_f32", 64, d_stamps, d_sink);
    run<1>("v_pk_fma_f32", 64, d_stamps, d_sink);
    run<2>("v_pk_mul_f32", 64, d_stamps, d_sink);
    run<3>("v_pk_add_f32", 64, d_stamps, d_sink);
    run<11>("v_pk_mul_f32 (SGPR pair operand)", 64, d_stamps, d_sink);
    run<4>("v_add_f32", 64, d_stamps, d_sink);
    run<5>("v_max_f32", 64, d_stamps, d_sink);
    run<6>("v_cvt_f32_ubyte1", 64, d_stamps, d_sink);
    run<7>("v_rcp_f32", 64, d_stamps, d_sink);
    run<8>("v_mov_b64", 64, d_stamps, d_sink);
    run<9>("v_cmp_lt_f32 + v_cndmask_b32 (pair)", 32, d_stamps, d_sink);
    run<10>("MOG fast-step mix (36 pk/26/2 rcp)", 64, d_stamps, d_sink);
    run<12>("v_mul_f32", 64, d_stamps, d_sink);
    run<13>("v_cndmask_b32 (vcc)", 64, d_stamps, d_sink);
    run<14>("v_cmp_lt_f32 (vcc)", 64, d_stamps, d_sink);
    run<15>("v_pk_fma_f32 + v_add_f32 (pair)", 32, d_stamps, d_sink);
    run<16>("v_pk_fma_f32 + v_max_f32 (pair)", 32, d_stamps, d_sink);
    run<17>("2 x v_fma_f32 (pair)", 32, d_stamps, d_sink);
    run<18>("s_add_u32", 64, d_stamps, d_sink);
    run<19>("s_nop 0", 64, d_stamps, d_sink);
    run<20>("v_pk_fma_f32 + s_add_u32 (pair)", 32, d_stamps, d_sink);
    run<21>("v_pk_fma_f32 + 2 s_add_u32 (triple)", 32, d_stamps, d_sink);
    run<22>("v_add_f32 + s_add_u32 (pair)", 32, d_stamps, d_sink);
    run<23>("s_cmp + s_cbranch not taken (pair)", 32, d_stamps, d_sink);
    return 0;
}
