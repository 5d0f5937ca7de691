// Microbenchmark (not part of the product): the memory access shapes of
// mog_batch2_k in isolation, on gfx950.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/hbm_shapes tools/ubench/hbm_shapes.hip
//   tools/ubench/hbm_shapes <case> [frames]          (on the GPU box)
//
// 1. Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for these shapes
//    (MI355X_MICROARCH.md: the counters are exact for 16 B/lane streaming
//    stores, read half for 16 B/lane streaming loads, "other access widths
//    are uncalibrated").  Every case moves a KNOWN number of bytes exactly
//    once over a buffer of more than 1 GB; run each under
//        rocprofv3 --pmc FETCH_SIZE -- tools/ubench/hbm_shapes <case>
//        rocprofv3 --pmc WRITE_SIZE -- tools/ubench/hbm_shapes <case>
//    and divide the counter by the printed byte count.
//      lines16   16 B per lane, whole 1-KiB runs per wave (the guide's shape)
//      seg8      the MOG frame read: 1080p BGR frames, a wave = a 32 x 4
//                pixel tile = four 96-byte row segments per frame, 8 B per
//                lane at a 6-byte stride (odd lanes 2 bytes early), every
//                frame of every tile once
//      seg12     the same bytes as 12 B per lane (32 lanes per frame)
//      plane8r   the MOG model read: float2 per lane, 128-byte rows of the
//                tile, 12 planes
//      plane8w   the MOG model write-back: float2 stores, same addresses
//      lines16w  16 B per lane streaming stores
// 2. Prefetch-depth skeleton: the frame loop of mog_batch2_k with the model
//    arithmetic replaced by a chain of NV packed FMAs per frame, at the
//    kernel's occupancy (64-thread workgroups, 4 waves per SIMD, 9 KB of LDS
//    per wave), frames requested DEPTH ahead through a rolling register ring
//    (case ring<DEPTH>) or through LDS with 12-byte LDS-DMA loads, two frames
//    per instruction (case dma<PAIRS in flight>).  Prints ms per 120 frames.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CHECK(x)                                                              \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

constexpr int W = 1920, H = 1080;
constexpr long NPIX = (long)W * H;
constexpr long FRAME_BYTES = NPIX * 3;
constexpr int N_GROUPS = (int)((NPIX / 2 + 255) / 256);
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// mog_batch2_k's block -> tile mapping: byte offset of the lane's pixel pair
// inside a frame (pixel index i0), or -1
__device__ __forceinline__ long tile_pixel(int lane) {
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const unsigned pos = (j >> 2) * 8 + xcd;
    if (pos >= (unsigned)N_GROUPS) return -1;
    const long g = ((long)pos * 4 + (j & 3)) * 64 + lane;
    if (g * 2 >= NPIX) return -1;
    const long tile = g >> 6;
    const int tiles_x = W >> 5;
    const long ty = tile / tiles_x;
    const int tx = (int)(tile - ty * tiles_x);
    return (ty * 4 + (lane >> 4)) * W + tx * 32 + (lane & 15) * 2;
}

// ---- calibration kernels ---------------------------------------------------
__global__ __launch_bounds__(256) void lines16_k(const u4 *src, long n16,
                                                 unsigned *sink) {
    unsigned acc = 0;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const u4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void lines16w_k(u4 *dst, long n16) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
        dst[i] = u4{(unsigned)i, 1u, 2u, 3u};
}

__global__ __launch_bounds__(64) void seg8_k(const uint8_t *bgr, int nframes,
                                             unsigned *sink) {
    const int lane = threadIdx.x;
    const long i0 = tile_pixel(lane);
    if (i0 < 0) return;
    const unsigned sh = (lane & 1) * 2u;
    const int off = (int)(i0 * 3) - (int)sh;
    unsigned acc = 0;
    for (int t0 = 0; t0 < nframes; t0 += 4) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(bgr + (long)t0 * FRAME_BYTES), 0, -1,
            0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u2 v = __builtin_bit_cast(
                u2, __builtin_amdgcn_raw_buffer_load_b64(
                        rsrc, off, (int)(j * FRAME_BYTES), 0));
            acc += v.x ^ v.y;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(64) void seg12_k(const uint8_t *bgr, int nframes,
                                              unsigned *sink) {
    // lanes 0-31: frame t, lanes 32-63: frame t + 1; lane l of a half reads
    // 12 bytes: row l / 8 of the tile, bytes (l % 8) * 12 of its segment
    const int lane = threadIdx.x;
    const long i0 = tile_pixel(0);
    if (i0 < 0) return;
    const int l = lane & 31;
    const int off = (int)(i0 * 3) + (l >> 3) * W * 3 + (l & 7) * 12 +
                    (lane >> 5) * (int)FRAME_BYTES;
    unsigned acc = 0;
    for (int t0 = 0; t0 < nframes; t0 += 8) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(bgr + (long)t0 * FRAME_BYTES), 0, -1,
            0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u3 v = __builtin_bit_cast(
                u3, __builtin_amdgcn_raw_buffer_load_b96(
                        rsrc, off, (int)(2 * j * FRAME_BYTES), 0));
            acc += v.x ^ v.y ^ v.z;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(64) void plane8r_k(const float *planes,
                                                int nplanes, unsigned *sink) {
    const long i0 = tile_pixel(threadIdx.x);
    if (i0 < 0) return;
    float acc = 0.f;
    for (int p = 0; p < nplanes; ++p) {
        const float2 v =
            *reinterpret_cast<const float2 *>(planes + (long)p * NPIX + i0);
        acc += v.x + v.y;
    }
    if (acc == 1234.5f) *sink = 1;
}

__global__ __launch_bounds__(64) void plane8w_k(float *planes, int nplanes) {
    const long i0 = tile_pixel(threadIdx.x);
    if (i0 < 0) return;
    for (int p = 0; p < nplanes; ++p)
        *reinterpret_cast<float2 *>(planes + (long)p * NPIX + i0) =
            make_float2((float)p, (float)i0);
}

// ---- prefetch-depth skeleton ---------------------------------------------
// NV packed FMAs per frame on 3 independent chains fed by the frame's bytes
// (so that the load cannot be dropped and the chain is about as serial as
// the real update)
template <int NV>
__device__ __forceinline__ void fake_update(f2 (&m)[3], unsigned w0,
                                            unsigned w1, float alpha) {
    f2 x0, x1, x2;
    x0.x = (float)(w0 & 255u);
    x1.x = (float)((w0 >> 8) & 255u);
    x2.x = (float)((w0 >> 16) & 255u);
    x0.y = (float)(w0 >> 24);
    x1.y = (float)(w1 & 255u);
    x2.y = (float)((w1 >> 8) & 255u);
    const f2 a = {alpha, alpha};
#pragma unroll
    for (int i = 0; i < (NV - 6) / 3; ++i) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(m[0]) : "v"(a), "v"(x0));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(m[1]) : "v"(a), "v"(x1));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(m[2]) : "v"(a), "v"(x2));
    }
}

constexpr int HOLD_LDS = 9216;      // what mog_batch2_k's cold model takes

template <int DEPTH, int NV, int WAVES>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void ring_k(
    const uint8_t *bgr, int nframes, float alpha, float *out) {
    __shared__ float hold[HOLD_LDS / 4];
    const int lane = threadIdx.x;
    const long i0 = tile_pixel(lane);
    if (i0 < 0) return;
    if (alpha == 77.f) hold[lane] = alpha;
    const unsigned sh = (lane & 1) * 16u;
    const int off = (int)(i0 * 3) - (int)(sh >> 3);
    f2 m[3] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}};
    unsigned ring[DEPTH][2];
    auto rsrc_of = [&](int t) {
        if (t + DEPTH > nframes) t = nframes - DEPTH;
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(bgr + (long)t * FRAME_BYTES), 0, -1,
            0x00020000);
    };
    {
        const __amdgpu_buffer_rsrc_t r0 = rsrc_of(0);
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const u2 v = __builtin_bit_cast(
                u2, __builtin_amdgcn_raw_buffer_load_b64(
                        r0, off, (int)(j * FRAME_BYTES), 0));
            ring[j][0] = v.x;
            ring[j][1] = v.y;
        }
    }
    for (int t0 = 0; t0 < nframes; t0 += DEPTH) {
        const __amdgpu_buffer_rsrc_t rn = rsrc_of(t0 + DEPTH);
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            // (asm pins the point where the frame's registers are first
            // read: left to itself the compiler decodes the whole ring at the
            // top of the iteration, behind a wait for every load in flight)
            unsigned c01, c2;
            asm volatile("v_alignbit_b32 %0, %2, %3, %4\n\t"
                         "v_lshrrev_b32 %1, %4, %2"
                         : "=&v"(c01), "=&v"(c2)
                         : "v"(ring[j][1]), "v"(ring[j][0]), "v"(sh));
            // the slot is free: request the frame DEPTH ahead into it
            const u2 v = __builtin_bit_cast(
                u2, __builtin_amdgcn_raw_buffer_load_b64(
                        rn, off, (int)(j * FRAME_BYTES), 0));
            ring[j][0] = v.x;
            ring[j][1] = v.y;
            fake_update<NV>(m, c01, c2, alpha);
        }
    }
    if (m[0].x + m[1].y + m[2].x == 1234.5f) out[0] = hold[lane];
}

// Frames through LDS: a ring of SLOTS pair-slots of 768 bytes (two frames of
// the tile), filled by one 12-byte-per-lane LDS-DMA load each.  No landing
// registers; the wave counts its own loads (vmcnt).
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

__device__ __forceinline__ void dma12(const uint8_t *base, unsigned voff,
                                      unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx3 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_addr)
        : "memory");
}

template <int SLOTS, int NV, int WAVES, int COLD>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void dma_k(
    const uint8_t *bgr, int nframes, float alpha, float *out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[COLD + SLOTS * 768];
    const int lane = threadIdx.x;
    const long i0 = tile_pixel(lane);
    if (i0 < 0) return;
    if (alpha == 77.f) lds[lane] = 1;
    const long t0px = tile_pixel(0);
    const int l = lane & 31;
    // source: this lane's 12 bytes of (frame pair base) + half * frame
    const unsigned voff = (unsigned)(t0px * 3) + (l >> 3) * W * 3 + (l & 7) * 12 +
                          (lane >> 5) * (unsigned)FRAME_BYTES;
    const unsigned ring0 =
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(lds + COLD);
    // reader: lane's pair inside a frame image of 4 x 96 bytes
    const unsigned sh = (lane & 1) * 16u;
    const unsigned rd = (lane >> 4) * 96 + (lane & 15) * 6 - (sh >> 3);
    f2 m[3] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}};
    const int npairs = nframes / 2;
    auto issue = [&](int p) {
        const int q = p < npairs ? p : npairs - 1;      // tail: re-read
        dma12(bgr + (long)q * 2 * FRAME_BYTES, voff,
              ring0 + (unsigned)(p % SLOTS) * 768u);
    };
#pragma unroll
    for (int p = 0; p < SLOTS - 1; ++p) issue(p);
    for (int p = 0; p < npairs; ++p) {
        issue(p + SLOTS - 1);
        wait_vm<SLOTS - 1>();
        const unsigned char *img = lds + COLD + (p % SLOTS) * 768;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const unsigned *w =
                reinterpret_cast<const unsigned *>(img + f * 384 + rd);
            const unsigned a = w[0], b = w[1];
            const unsigned c01 = __builtin_amdgcn_alignbit(b, a, sh);
            const unsigned c2 = b >> sh;
            fake_update<NV>(m, c01, c2, alpha);
        }
        // the slot is re-filled by the issue() of the next iteration: the
        // reads above must have returned before that (they feed VALU work
        // that precedes it in program order; lgkmcnt is the compiler's)
    }
    wait_vm<0>();
    if (m[0].x + m[1].y + m[2].x == 1234.5f) out[0] = (float)lds[lane];
}

// Where the bytes of a 12-byte-per-lane LDS-DMA load land: lane l reads
// dwords 3l, 3l+1, 3l+2 of a table whose dword i holds i; the LDS image is
// copied out.
__global__ __launch_bounds__(64) void dma12_layout_k(const unsigned *table,
                                                     unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned img[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) img[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned lds =
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)img;
    dma12(reinterpret_cast<const uint8_t *>(table), (unsigned)lane * 12u, lds);
    wait_vm<0>();
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = img[i];
}

// ---------------------------------------------------------------------------
static float timed(void (*launch)(), int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static uint8_t *g_bgr;
static float *g_planes, *g_out;
static unsigned *g_sink;
static int g_frames = 180;
static const unsigned BLOCKS = (unsigned)((N_GROUPS + 7) / 8) * 8 * 4;

#define RING_CASE(D, NV, WV)                                                  \
    if (which == "ring" #D "_" #NV "_" #WV) {                                 \
        const float ms = timed(                                               \
            [] {                                                              \
                hipLaunchKernelGGL((ring_k<D, NV, WV>), dim3(BLOCKS),         \
                                   dim3(64), 0, 0, g_bgr, 120, 0.005f,        \
                                   g_out);                                    \
            },                                                                \
            10);                                                              \
        printf("%-16s %.4f ms per 120 frames  (%.2f TB/s of frame bytes)\n",  \
               which.c_str(), ms, 120 * (double)FRAME_BYTES / ms / 1e9);      \
        return 0;                                                             \
    }
#define DMA_CASE(S, NV, WV, COLD)                                             \
    if (which == "dma" #S "_" #NV "_" #WV "_" #COLD) {                        \
        const float ms = timed(                                               \
            [] {                                                              \
                hipLaunchKernelGGL((dma_k<S, NV, WV, COLD>), dim3(BLOCKS),    \
                                   dim3(64), 0, 0, g_bgr, 120, 0.005f,        \
                                   g_out);                                    \
            },                                                                \
            10);                                                              \
        printf("%-16s %.4f ms per 120 frames  (%.2f TB/s of frame bytes)\n",  \
               which.c_str(), ms, 120 * (double)FRAME_BYTES / ms / 1e9);      \
        return 0;                                                             \
    }

static int run(const std::string &which) {
    if (which == "lines16") {
        const long n16 = (long)g_frames * FRAME_BYTES / 16;
        hipLaunchKernelGGL(lines16_k, dim3(256 * 16), dim3(256), 0, 0,
                           (const u4 *)g_bgr, n16, g_sink);
        CHECK(hipDeviceSynchronize());
        printf("lines16  bytes_read %ld\n", n16 * 16);
        return 0;
    }
    if (which == "lines16w") {
        const long n16 = (long)g_frames * FRAME_BYTES / 16;
        hipLaunchKernelGGL(lines16w_k, dim3(256 * 16), dim3(256), 0, 0,
                           (u4 *)g_bgr, n16);
        CHECK(hipDeviceSynchronize());
        printf("lines16w bytes_written %ld\n", n16 * 16);
        return 0;
    }
    if (which == "seg8") {
        hipLaunchKernelGGL(seg8_k, dim3(BLOCKS), dim3(64), 0, 0, g_bgr,
                           g_frames, g_sink);
        CHECK(hipDeviceSynchronize());
        printf("seg8     bytes_read %ld (useful; the lanes ask for %ld)\n",
               (long)g_frames * FRAME_BYTES, (long)g_frames * NPIX / 2 * 8);
        return 0;
    }
    if (which == "seg12") {
        hipLaunchKernelGGL(seg12_k, dim3(BLOCKS), dim3(64), 0, 0, g_bgr,
                           g_frames, g_sink);
        CHECK(hipDeviceSynchronize());
        printf("seg12    bytes_read %ld\n", (long)g_frames * FRAME_BYTES);
        return 0;
    }
    if (which == "plane8r") {
        hipLaunchKernelGGL(plane8r_k, dim3(BLOCKS), dim3(64), 0, 0, g_planes,
                           40, g_sink);
        CHECK(hipDeviceSynchronize());
        printf("plane8r  bytes_read %ld\n", 40 * NPIX * 4);
        return 0;
    }
    if (which == "dma12layout") {
        unsigned *tab, *out, host[512];
        CHECK(hipMalloc((void **)&tab, 4096));
        CHECK(hipMalloc((void **)&out, 4096));
        for (int i = 0; i < 512; ++i) host[i] = i;
        CHECK(hipMemcpy(tab, host, 2048, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(dma12_layout_k, dim3(1), dim3(64), 0, 0, tab, out);
        CHECK(hipMemcpy(host, out, 2048, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; ++i)
            printf("%s%d", i % 16 ? " " : "\n", (int)host[i]);
        printf("\n");
        return 0;
    }
    if (which == "plane8w") {
        hipLaunchKernelGGL(plane8w_k, dim3(BLOCKS), dim3(64), 0, 0, g_planes,
                           40);
        CHECK(hipDeviceSynchronize());
        printf("plane8w  bytes_written %ld\n", 40 * NPIX * 4);
        return 0;
    }
    RING_CASE(4, 46, 4) RING_CASE(8, 46, 4) RING_CASE(12, 46, 4)
    RING_CASE(16, 46, 4) RING_CASE(4, 76, 4) RING_CASE(8, 76, 4)
    RING_CASE(16, 76, 4) RING_CASE(4, 12, 4) RING_CASE(16, 12, 4)
    RING_CASE(8, 46, 3) RING_CASE(16, 46, 3)
    DMA_CASE(2, 46, 4, 9216) DMA_CASE(5, 46, 4, 6144) DMA_CASE(8, 46, 4, 3072)
    DMA_CASE(8, 76, 4, 3072) DMA_CASE(5, 76, 4, 6144) DMA_CASE(8, 12, 4, 3072)
    DMA_CASE(12, 46, 3, 3072) DMA_CASE(3, 46, 4, 6144)
    fprintf(stderr, "unknown case %s\n", which.c_str());
    return 2;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: hbm_shapes <case> [frames]\n");
        return 2;
    }
    if (argc > 2) g_frames = atoi(argv[2]);
    if (g_frames < 120) g_frames = 120;
    g_frames &= ~7;
    CHECK(hipMalloc((void **)&g_bgr, (size_t)g_frames * FRAME_BYTES + 4096));
    CHECK(hipMemset(g_bgr, 0x5a, (size_t)g_frames * FRAME_BYTES + 4096));
    CHECK(hipMalloc((void **)&g_planes, sizeof(float) * 40 * NPIX));
    CHECK(hipMemset(g_planes, 0, sizeof(float) * 40 * NPIX));
    CHECK(hipMalloc((void **)&g_out, 4096));
    CHECK(hipMalloc((void **)&g_sink, 4096));
    // flush what the memsets left in the caches: an unrelated 512 MiB sweep
    {
        void *junk = nullptr;
        CHECK(hipMalloc(&junk, 512u << 20));
        CHECK(hipMemset(junk, 1, 512u << 20));
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(junk));
    }
    if (std::string(argv[1]) == "all") {
        const char *cases[] = {
            "ring4_12_4", "ring16_12_4", "ring4_46_4", "ring8_46_4",
            "ring12_46_4", "ring16_46_4", "ring8_46_3", "ring16_46_3",
            "ring4_76_4", "ring8_76_4", "ring16_76_4", "dma2_46_4_9216",
            "dma3_46_4_6144", "dma5_46_4_6144", "dma8_46_4_3072",
            "dma12_46_3_3072", "dma5_76_4_6144", "dma8_76_4_3072",
            "dma8_12_4_3072"};
        for (const char *c : cases) run(c);
        return 0;
    }
    return run(argv[1]);
}
