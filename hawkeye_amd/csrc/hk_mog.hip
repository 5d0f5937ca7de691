// K1 -- MOG background subtraction for gfx950.
//
// Replaces cv2.bgsegm.createBackgroundSubtractorMOG().apply(frame)
// (reference tracker.py:357,395; arithmetic in opencv_contrib bgsegm
// bgfg_gaussmix.cpp process8uC3, restated in oracle/detect_ref.c).
//
// Layout: the per-pixel mixture (5 components x {weight, sortKey, mean[3],
// var[3]}) is stored as 40 fp32 planes of N = W*H values (SoA), so that a
// wave's 64 lanes touch 64 consecutive floats (or float4s) of one plane.
// Plane p = field*5 + k, fields in the order of enum Field.
//
// The recursion is independent per pixel, so one launch advances a pixel
// through many frames with its mixture held on chip: per frame only the 3
// input bytes are read and the mask byte written.  hk_mog_apply is the same
// kernel with one frame.  Pixels are handled in pairs: the common case (both
// match their first component, the background mode) runs in packed fp32
// (v_pk_*), anything else through the general per-pixel step.
//
// Arithmetic is bit-identical to the oracle: same fp32 operation order, no
// FMA contraction (-ffp-contract=off), IEEE divide and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved.

#include <cfloat>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "hk_common.h"

namespace {

constexpr int K = 5;
constexpr int NPLANES = 8 * K;
constexpr int HISTORY = 200;
// learning rates 1 / min(n, HISTORY) for n = 1 .., padded so that the batch
// kernel can fetch the rates of a group of 4 frames from one clamped index
constexpr int ALPHA_ENTRIES = HISTORY + 4;

enum Field { F_W = 0, F_SK, F_M0, F_M1, F_M2, F_V0, F_V1, F_V2 };

// OpenCV's defaults, computed exactly as bgfg_gaussmix.cpp computes them.
struct MogConst {
    float T, vT, w0, sk0, var0, minVar;
};

struct Mix {                // one pixel's mixture, lives in VGPRs
    float f[8][K];          // [field][component]; every index is a
                            // compile-time constant after unrolling
    __device__ __forceinline__ float get(int q, int k) const { return f[q][k]; }
    __device__ __forceinline__ void set(int q, int k, float v) { f[q][k] = v; }
};

typedef float f2 __attribute__((ext_vector_type(2)));

// Two neighbouring pixels: .x and .y of every value sit in an aligned VGPR
// pair (v_pk_* ops).  Only what every frame touches lives in registers: all
// weights and sort keys, and the mean / variance of components 0 and 1.  The
// mean / variance of components 2-4 ("cold": read only when a pixel leaves
// its first two modes) live in LDS, 18 values x 2 pixels per thread, so that
// the kernel fits 128 VGPRs without spilling to scratch.
constexpr int COLD_FROM = 2;
constexpr int N_COLD = 6 * (K - COLD_FROM);         // 18
__device__ __forceinline__ constexpr bool is_cold(int q, int k) {
    return k >= COLD_FROM && q >= F_M0;
}
__device__ __forceinline__ constexpr int cold_slot(int q, int k) {
    return ((q - F_M0) * (K - COLD_FROM) + (k - COLD_FROM)) * 128;
}

struct Mix2 {
    f2 f[8][K];             // cold entries are never touched (dead registers)
    float *cold;            // this thread's LDS base: cold[slot + pixel]
};

template <int L>
struct Lane {               // one pixel of a Mix2, seen as a Mix
    Mix2 &m;
    __device__ __forceinline__ float get(int q, int k) const {
        if (is_cold(q, k)) return m.cold[cold_slot(q, k) + L];
        return L ? m.f[q][k].y : m.f[q][k].x;
    }
    __device__ __forceinline__ void set(int q, int k, float v) {
        if (is_cold(q, k)) m.cold[cold_slot(q, k) + L] = v;
        else if (L) m.f[q][k].y = v;
        else m.f[q][k].x = v;
    }
};

__device__ __forceinline__ float max_like_std(float a, float b) {
    return (a < b) ? b : a;             // std::max(a, b)
}

// One frame of one pixel, the general case.  Returns 255 (foreground) or 0.
template <class P>
__device__ __forceinline__ unsigned mog_step(P &p, float x0, float x1,
                                             float x2, float alpha,
                                             const MogConst c) {
    float wsum = 0.f;
    int kHit = -1;
    int kEnd = K;           // k at which the scan stopped
    bool scanning = true;

#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (scanning) {
            const float w = p.get(F_W, k);
            wsum += w;
            if (w < FLT_EPSILON) {
                scanning = false;
                kEnd = k;
            } else {
                const float d0 = x0 - p.get(F_M0, k);
                const float d1 = x1 - p.get(F_M1, k);
                const float d2 = x2 - p.get(F_M2, k);
                float v0 = p.get(F_V0, k), v1 = p.get(F_V1, k),
                      v2 = p.get(F_V2, k);
                float dist2 = d0 * d0;
                dist2 = dist2 + d1 * d1;
                dist2 = dist2 + d2 * d2;
                if (dist2 < c.vT * (v0 + v1 + v2)) {
                    wsum -= w;
                    const float dw = alpha * (1.f - w);
                    p.set(F_W, k, w + dw);
                    p.set(F_M0, k, p.get(F_M0, k) + alpha * d0);
                    p.set(F_M1, k, p.get(F_M1, k) + alpha * d1);
                    p.set(F_M2, k, p.get(F_M2, k) + alpha * d2);
                    v0 = max_like_std(v0 + alpha * (d0 * d0 - v0), c.minVar);
                    v1 = max_like_std(v1 + alpha * (d1 * d1 - v1), c.minVar);
                    v2 = max_like_std(v2 + alpha * (d2 * d2 - v2), c.minVar);
                    p.set(F_V0, k, v0);
                    p.set(F_V1, k, v1);
                    p.set(F_V2, k, v2);
                    p.set(F_SK, k, w / sqrtf(v0 + v1 + v2));

                    // keep components ordered by sortKey (insertion step)
                    int pos = k;
                    bool moving = true;
#pragma unroll
                    for (int k1 = k - 1; k1 >= 0; --k1) {
                        if (moving) {
                            if (p.get(F_SK, k1) >= p.get(F_SK, k1 + 1)) {
                                moving = false;
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    const float t = p.get(q, k1);
                                    p.set(q, k1, p.get(q, k1 + 1));
                                    p.set(q, k1 + 1, t);
                                }
                                pos = k1;
                            }
                        }
                    }
                    kHit = pos;
                    // upstream adds the tail from index k of the REORDERED
                    // array; reproduced as is
#pragma unroll
                    for (int kk = k; kk < K; ++kk) wsum += p.get(F_W, kk);
                    scanning = false;
                    kEnd = k;
                }
            }
        }
    }

    if (kHit < 0) {         // nothing matched: new component in slot kr
        const int kr = kEnd < K - 1 ? kEnd : K - 1;
        kHit = kr;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            if (kk == kr) {
                wsum += c.w0 - p.get(F_W, kk);
                p.set(F_W, kk, c.w0);
                p.set(F_M0, kk, x0);
                p.set(F_M1, kk, x1);
                p.set(F_M2, kk, x2);
                p.set(F_V0, kk, c.var0);
                p.set(F_V1, kk, c.var0);
                p.set(F_V2, kk, c.var0);
                p.set(F_SK, kk, c.sk0);
            }
        }
    }

    const float wscale = 1.f / wsum;
    wsum = 0.f;
    int kFg = -1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        p.set(F_W, k, p.get(F_W, k) * wscale);
        wsum += p.get(F_W, k);
        p.set(F_SK, k, p.get(F_SK, k) * wscale);
        if (wsum > c.T && kFg < 0) kFg = k + 1;
    }
    return (kHit >= kFg) ? 255u : 0u;
}

// ---------------------------------------------------------------------------
// The common case -- both pixels of a lane match their component 0 (the
// background mode) -- in packed fp32: every add/mul/fma handles the two pixels
// at once.  Same operations in the same order as mog_step, with these
// value-preserving shortcuts (the operands' ranges are known here):
//  * k == 0: wsum = ((0 + w) - w) + W0' + W1.. = W0' + W1.. exactly;
//  * sqrt / divide: the compiler's correctly rounded sequences minus their
//    range scaling and special-case fix-ups, which are the identity for
//    12.9 <= sqrt argument <= 2e5, FLT_EPSILON <= w <= 1, 2^-23 <= wsum <= 8
//    (hk_mog_selfcheck_math compares them with the IEEE operators over these
//    ranges);
//  * max(v, minVar) as v_max_f32 (v is never NaN);
//  * kHit == 0: foreground iff the weights never sum past T (cumulative sums
//    of non-negative weights are monotone, so kHit >= kFg reduces to that).
// ---------------------------------------------------------------------------
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) {
    return __builtin_elementwise_fma(a, b, c);
}

// correctly rounded sqrt for normal x with no range scaling needed
__device__ __forceinline__ float sqrt_ranged(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1);
    const float sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    float r = (0.f >= rm) ? sm : s;
    r = (0.f < rp) ? sp : r;
    return r;
}

__device__ __forceinline__ f2 sqrt_ranged2(f2 x) {
    f2 s, sm, sp;
    s.x = __builtin_amdgcn_sqrtf(x.x);
    s.y = __builtin_amdgcn_sqrtf(x.y);
    sm.x = __int_as_float(__float_as_int(s.x) - 1);
    sm.y = __int_as_float(__float_as_int(s.y) - 1);
    sp.x = __int_as_float(__float_as_int(s.x) + 1);
    sp.y = __int_as_float(__float_as_int(s.y) + 1);
    const f2 rm = fma2(-sm, s, x);
    const f2 rp = fma2(-sp, s, x);
    f2 r;
    r.x = (0.f >= rm.x) ? sm.x : s.x;
    r.y = (0.f >= rm.y) ? sm.y : s.y;
    r.x = (0.f < rp.x) ? sp.x : r.x;
    r.y = (0.f < rp.y) ? sp.y : r.y;
    return r;
}

// correctly rounded n / d when no operand or intermediate needs scaling
__device__ __forceinline__ float div_ranged(float n, float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.f);
    const float r = __builtin_fmaf(e0, r0, r0);
    const float q0 = n * r;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r, q1);
}

__device__ __forceinline__ f2 div_ranged2(f2 n, f2 d) {
    f2 r0;
    r0.x = __builtin_amdgcn_rcpf(d.x);
    r0.y = __builtin_amdgcn_rcpf(d.y);
    const f2 one = {1.f, 1.f};
    const f2 e0 = fma2(-d, r0, one);
    const f2 r = fma2(e0, r0, r0);
    const f2 q0 = n * r;
    const f2 e1 = fma2(-d, q0, n);
    const f2 q1 = fma2(e1, r, q0);
    const f2 e2 = fma2(-d, q1, n);
    return fma2(e2, r, q1);
}

// sortKey of component 0, deferred.  A fast step would set
//     SK[0] = (w_old / sqrt(var sum)) * wscale,
// a value nothing reads if the next frame takes the fast step again (it is
// overwritten from scratch).  So the fast step only notes the operands; the
// square root, the division and the product are done when the value is
// needed: before a general step of this lane, and before the model is stored.
// The variance sum is the lane's carried `vsum` (see the kernel): it does not
// change between the fast step and the moment the key is materialised.
struct LazySK {
    f2 w, scale;            // w.x == 0: SK[0] in the model is current (the
                            // two pixels of a lane take the fast step
                            // together, and a weight that does is > 0)
};

__device__ __forceinline__ void materialise_sk0(Mix2 &p, LazySK &z, f2 vsum) {
    const bool pend = z.w.x != 0.f;
    f2 s, w;
    s.x = pend ? vsum.x : 1024.f;
    s.y = pend ? vsum.y : 1024.f;
    w.x = pend ? z.w.x : 1.f;
    w.y = pend ? z.w.y : 1.f;
    const f2 sk = div_ranged2(w, sqrt_ranged2(s)) * z.scale;
    p.f[F_SK][0].x = pend ? sk.x : p.f[F_SK][0].x;
    p.f[F_SK][0].y = pend ? sk.y : p.f[F_SK][0].y;
    z.w.x = 0.f;
}

// OpenCV's constants as literals for the fast step (a kernel argument is not
// known to be free of NaNs, which costs a canonicalising instruction per use);
// hk_mog_create checks that opencv_defaults() yields exactly these.
constexpr float FAST_T = 0.7f, FAST_VT = 6.25f, FAST_MINVAR = 225.f;

// Both pixels of a lane against their component 0: differences, their
// squares and the squared distance.
struct Probe {
    f2 d0, d1, d2, q0, q1, q2, dist2;
};

__device__ __forceinline__ Probe probe0(const Mix2 &p, f2 x0, f2 x1, f2 x2) {
    Probe r;
    r.d0 = x0 - p.f[F_M0][0];
    r.d1 = x1 - p.f[F_M1][0];
    r.d2 = x2 - p.f[F_M2][0];
    r.q0 = r.d0 * r.d0;
    r.q1 = r.d1 * r.d1;
    r.q2 = r.d2 * r.d2;
    r.dist2 = r.q0 + r.q1;
    r.dist2 = r.dist2 + r.q2;
    return r;
}

// The fast step's update of a wave in which every pixel has one component
// whose weight is exactly 1 (a pixel that has only ever seen its background:
// the steady state of most of the picture).  Then
//     wn = 1 + alpha * (1 - 1) = 1,  wsum = 1,  wscale = 1 / 1 = 1,
//     W[0] = wn * wscale = 1,        cum = 1 > T: background,
// all exactly, so the weight arithmetic of weighted_update is skipped whole:
// only the mean and the variance move.  The deferred sort key's operands are
// w = 1 and wscale = 1; the kernel notes that once for the wave (mode -1)
// instead of writing them into `z` every frame.
__device__ __forceinline__ void unit_update(Mix2 &p, f2 &vsum, const Probe &r,
                                            float alpha) {
    f2 v0 = p.f[F_V0][0], v1 = p.f[F_V1][0], v2 = p.f[F_V2][0];
    p.f[F_M0][0] = p.f[F_M0][0] + alpha * r.d0;
    p.f[F_M1][0] = p.f[F_M1][0] + alpha * r.d1;
    p.f[F_M2][0] = p.f[F_M2][0] + alpha * r.d2;
    v0 = v0 + alpha * (r.q0 - v0);
    v1 = v1 + alpha * (r.q1 - v1);
    v2 = v2 + alpha * (r.q2 - v2);
    v0.x = __builtin_fmaxf(v0.x, FAST_MINVAR);
    v0.y = __builtin_fmaxf(v0.y, FAST_MINVAR);
    v1.x = __builtin_fmaxf(v1.x, FAST_MINVAR);
    v1.y = __builtin_fmaxf(v1.y, FAST_MINVAR);
    v2.x = __builtin_fmaxf(v2.x, FAST_MINVAR);
    v2.y = __builtin_fmaxf(v2.y, FAST_MINVAR);
    p.f[F_V0][0] = v0;
    p.f[F_V1][0] = v1;
    p.f[F_V2][0] = v2;
    vsum = (v0 + v1) + v2;
}

// The fast step's update in a wave whose lanes hold at most nc components
// (wave-uniform; 1, 2 or K).  A component that was never created has weight 0
// and sort key 0: adding the one to a positive sum and scaling either changes
// nothing, so those operations are skipped -- same results.  Returns the
// cumulated weights (kHit == 0: a pixel is foreground iff its weights never
// sum past T); SK[0] is deferred into `z`.
__device__ __forceinline__ f2 weighted_update(Mix2 &p, LazySK &z, f2 &vsum,
                                              int nc, const Probe &pr,
                                              float alpha) {
    const f2 w = p.f[F_W][0];
    f2 v0 = p.f[F_V0][0], v1 = p.f[F_V1][0], v2 = p.f[F_V2][0];
    const f2 one = {1.f, 1.f};
    const f2 wn = w + alpha * (one - w);
    // (the empty asm statements keep these wave-uniform branches branches:
    // turned into selects they would cost more than they skip)
    f2 wsum = wn;
    if (nc >= 2) {
        asm volatile("; two components");
        wsum = wsum + p.f[F_W][1];
        if (nc > 2) {
            asm volatile("; up to five components");
#pragma unroll
            for (int k = 2; k < K; ++k) wsum = wsum + p.f[F_W][k];
        }
    }
    // 1 / wsum: its dependent fma chain is interleaved by hand with the
    // independent mean / variance updates (same operations, same operands)
    f2 r0;
    r0.x = __builtin_amdgcn_rcpf(wsum.x);
    r0.y = __builtin_amdgcn_rcpf(wsum.y);
    p.f[F_M0][0] = p.f[F_M0][0] + alpha * pr.d0;
    const f2 e0 = fma2(-wsum, r0, one);
    p.f[F_M1][0] = p.f[F_M1][0] + alpha * pr.d1;
    const f2 r = fma2(e0, r0, r0);
    p.f[F_M2][0] = p.f[F_M2][0] + alpha * pr.d2;
    const f2 e1 = fma2(-wsum, r, one);          // q0 = 1 * r = r
    v0 = v0 + alpha * (pr.q0 - v0);
    const f2 q1r = fma2(e1, r, r);
    v1 = v1 + alpha * (pr.q1 - v1);
    const f2 e2 = fma2(-wsum, q1r, one);
    v2 = v2 + alpha * (pr.q2 - v2);
    const f2 wscale = fma2(e2, r, q1r);
    v0.x = __builtin_fmaxf(v0.x, FAST_MINVAR);
    v0.y = __builtin_fmaxf(v0.y, FAST_MINVAR);
    v1.x = __builtin_fmaxf(v1.x, FAST_MINVAR);
    v1.y = __builtin_fmaxf(v1.y, FAST_MINVAR);
    v2.x = __builtin_fmaxf(v2.x, FAST_MINVAR);
    v2.y = __builtin_fmaxf(v2.y, FAST_MINVAR);
    p.f[F_V0][0] = v0;
    p.f[F_V1][0] = v1;
    p.f[F_V2][0] = v2;
    z.w = w;
    z.scale = wscale;
    vsum = (v0 + v1) + v2;

    p.f[F_W][0] = wn * wscale;
    f2 cum = p.f[F_W][0];
    if (nc >= 2) {
        asm volatile("; two components");
        p.f[F_W][1] = p.f[F_W][1] * wscale;
        cum = cum + p.f[F_W][1];
        p.f[F_SK][1] = p.f[F_SK][1] * wscale;
        if (nc > 2) {
            asm volatile("; up to five components");
#pragma unroll
            for (int k = 2; k < K; ++k) {
                p.f[F_W][k] = p.f[F_W][k] * wscale;
                cum = cum + p.f[F_W][k];
                p.f[F_SK][k] = p.f[F_SK][k] * wscale;
            }
        }
    }
    return cum;
}

// Lane `lane_sel` of a0 .. a3 set to the (wave-uniform) words w0 .. w3: four
// v_writelane_b32 with the lane number in M0 (a VOP3 takes one scalar
// register operand only, and the lane is not a compile-time constant).  M0 is
// saved and restored inside the statement.
__device__ __forceinline__ void write_lane_x4(unsigned &a0, unsigned &a1,
                                              unsigned &a2, unsigned &a3,
                                              int lane_sel, unsigned w0,
                                              unsigned w1, unsigned w2,
                                              unsigned w3) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %4, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "v_writelane_b32 %0, %5, m0\n\t"
        "v_writelane_b32 %1, %6, m0\n\t"
        "v_writelane_b32 %2, %7, m0\n\t"
        "v_writelane_b32 %3, %8, m0\n\t"
        "s_mov_b32 m0, %4"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(keep)
        : "s"(w0), "s"(w1), "s"(w2), "s"(w3), "s"(lane_sel));
}

// ---------------------------------------------------------------------------
// Fast path: contiguous frames, N % 2 == 0.  Each lane owns 2 neighbouring
// pixels (a 6-byte load per frame), the model is read once and written once
// per launch.
//
// What bounds it (MI355X, 1080p, 128 frames per launch; tools/ubench,
// profiles/): instruction issue -- vector AND scalar.  A scalar instruction
// (mask logic, compares, branches, address arithmetic, waits) takes a SIMD's
// issue slot for about as long as a vector one, so the frame step is written
// for few instructions of either kind on the path nearly every wave takes:
//   * mode <= 0 ("unit"): every pixel of the wave has one component of weight
//     exactly 1 and matches it: 26 packed + 14 plain vector instructions and a
//     handful of scalar ones per frame, no mask manipulation (unit_update);
//   * all lanes match their component 0 but weights are not trivial:
//     weighted_update, not masked either;
//   * otherwise lanes that match take weighted_update under their mask and the
//     others the general step (mog_step, the oracle's loop verbatim).
// Pixels off the first path come in compact 2-D groups (moving objects, their
// edges), hence the 32 x 4 pixel tiles per wave; such waves run several times
// longer than the others and there are only ~8 waves per slot in a launch,
// hence one wave per workgroup and the launch order (mog_order_k).  4 waves
// per SIMD need <= 128 VGPRs: the rarely used part of the model sits in LDS;
// DEPTH frames of input (and their learning rates) are requested ahead.
//
// DEPTH > 1 needs nframes % DEPTH == 0 (the host splits a ragged batch);
// alpha_table holds HISTORY + 4 entries, 1 / min(i + 1, HISTORY).
// MASKS: byte masks are written (hk_mog_apply_batch); TILES: the 1-bit tile
// words are (hk_mog_apply_batch_tiles, TILED geometry only).
// ---------------------------------------------------------------------------
// tools/probe build only (-DHK_MOG_PATH_STATS): how many wave-frames take
// which update path (tools/probe/mog_paths.py)
// tools/probe build only (-DHK_MOG_WAVE_TIMES): every wave of the batch
// kernel notes when it started and ended (s_memrealtime, 100 MHz) and how
// many of its frames left the fast step (tools/probe/mog_timeline.py)
#ifdef HK_MOG_WAVE_TIMES
__device__ unsigned long long *hk_mog_wave_times_dev;
// 0: the frame loads are dropped by the buffer range check (they return
// zeros: a black static scene) while every instruction stays -- what the
// kernel costs without its frame traffic
__device__ int hk_mog_frame_records_dev = -1;
#define HK_MOG_FRAME_RECORDS hk_mog_frame_records_dev
#else
#define HK_MOG_FRAME_RECORDS (-1)
#endif
#ifdef HK_MOG_PATH_STATS
__device__ unsigned long long hk_mog_path_stats_dev[8];
#define HK_PATH_STAT(i)                                                      \
    do {                                                                      \
        if (threadIdx.x == 0) atomicAdd(&hk_mog_path_stats_dev[i], 1ull);     \
    } while (0)
#else
#define HK_PATH_STAT(i) do {} while (0)
#endif

template <int WAVES, int DEPTH, bool TILED, bool MASKS, bool TILES>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void mog_batch2_k(
    float *__restrict__ planes, long n_pixels, int width,
    const uint8_t *__restrict__ bgr,
    long bgr_frame_stride, uint8_t *__restrict__ mask, long mask_frame_stride,
    int nframes, int frames_before, const float *__restrict__ alpha_table,
    const MogConst c, int n_groups, const int *__restrict__ order,
    int *__restrict__ cost, unsigned *__restrict__ tiles,
    long tile_words_per_frame) {
    static_assert(!TILES || TILED, "tile words need the tiled geometry");
    static_assert(DEPTH <= 4, "alpha_table is padded by 4 entries");
    // One wave per workgroup, so that a finished wave's slot and LDS are
    // reused at once (waves of one workgroup end at very different times).
    // A "group" is 4 units of 128 pixels that share cache lines of the frame
    // rows; workgroups go to the 8 XCDs round-robin by index, so the 4 units
    // of a group take indices 8 apart and meet in one XCD's L2.  `order`
    // (may be null) lists the groups, slowest of the previous launch first:
    // see mog_order_k.
#ifdef HK_MOG_WAVE_TIMES
    unsigned long long wave_t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(wave_t0)::"memory");
#endif
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const unsigned pos = (j >> 2) * 8 + xcd;
    if (pos >= (unsigned)n_groups) return;
    const unsigned group = order ? (unsigned)order[pos] : pos;
    const long g = ((long)group * 4 + (j & 3)) * 64 + threadIdx.x;  // pair
    long i0 = g * 2;
    if (i0 >= n_pixels) return;
    const int lane = threadIdx.x;
    if (TILED) {
        // A wave owns a 32 x 4 pixel tile instead of a 128-pixel run of a
        // row: pixels off the fast path come in compact 2-D groups (moving
        // objects, their edges), which then touch ~3x fewer waves.
        // Requires width % 32 == 0 and height % 4 == 0.
        const long tile = g >> 6;
        const int tiles_x = width >> 5;
        const long ty = tile / tiles_x;
        const int tx = (int)(tile - ty * tiles_x);
        i0 = (ty * 4 + (lane >> 4)) * width + tx * 32 + (lane & 15) * 2;
    }
    // TILES: the wave's tile takes 4 words per frame in the 1-bit output (see
    // hk_mog_apply_batch_tiles).  The words of 64 frames are collected in
    // four registers, lane L = frame L of the group of 64, and stored as one
    // 16-byte store per lane.  The registers start at zero and a frame without
    // foreground leaves its lane alone.
    unsigned *tile_dst = nullptr;
    unsigned ta0 = 0u, ta1 = 0u, ta2 = 0u, ta3 = 0u;
    if (TILES)
        tile_dst = tiles + (g >> 6) * 4 + (long)lane * tile_words_per_frame;

    // A component that was never created has weight 0 and all-zero fields
    // (the model starts zeroed, components are created with weight w0 > 0 and
    // weights are only ever rescaled), so its planes are neither read nor
    // written back: most pixels have one or two modes.
    __shared__ float cold_lds[N_COLD * 128];
    Mix2 px;
    px.cold = cold_lds + threadIdx.x * 2;
    float *const base = planes + i0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float2 w = *reinterpret_cast<const float2 *>(
            base + (long)(F_W * K + k) * n_pixels);
        px.f[F_W][k].x = w.x;
        px.f[F_W][k].y = w.y;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool used = k == 0 || px.f[F_W][k].x != 0.f ||
                          px.f[F_W][k].y != 0.f;
#pragma unroll
        for (int q = F_SK; q < 8; ++q) {
            float2 v = make_float2(0.f, 0.f);
            if (used)
                v = *reinterpret_cast<const float2 *>(
                    base + (long)(q * K + k) * n_pixels);
            if (is_cold(q, k)) {    // zeros too: one pixel of the pair may
                px.cold[cold_slot(q, k)] = v.x;         // create the component
                px.cold[cold_slot(q, k) + 1] = v.y;     // and both are stored
            } else {
                px.f[q][k].x = v.x;
                px.f[q][k].y = v.y;
            }
        }
    }

    LazySK lazy;
    lazy.w = lazy.scale = f2{0.f, 0.f};
    int slow_steps = 0;     // frames in which this wave left the fast step
    const unsigned long long all_lanes = __builtin_amdgcn_ballot_w64(true);
    // variance sum of component 0: the fast step's threshold operand, carried
    // from frame to frame (the value the test needs is the one the previous
    // step's update ended with) and recomputed after a general step
    f2 vsum = (px.f[F_V0][0] + px.f[F_V1][0]) + px.f[F_V2][0];
    // mode (wave-uniform): the largest number of components any lane holds
    // (1, 2, or K for "more"), or, if that is 1 and every W[0] is exactly 1
    // (see unit_update), 0; -1 = 0 and every lane's deferred sort key has the
    // operands w = 1, wscale = 1 (what unit_update leaves without writing it)
    auto wave_mode = [&]() {
        const bool two = px.f[F_W][1].x != 0.f || px.f[F_W][1].y != 0.f;
        const bool more = px.f[F_W][2].x != 0.f || px.f[F_W][2].y != 0.f;
        if (__builtin_amdgcn_ballot_w64(more)) return K;
        if (__builtin_amdgcn_ballot_w64(two)) return 2;
        return __builtin_amdgcn_ballot_w64(px.f[F_W][0].x != 1.f ||
                                           px.f[F_W][0].y != 1.f) ? 1 : 0;
    };
    int mode = wave_mode();

    // The 6 input bytes of a frame are far from those of the next frame, so
    // every load is an HBM miss (~2 us under load): the frames of the next
    // group of DEPTH are requested while this group is processed, and so are
    // their learning rates.
    uint8_t *dst = mask + i0;
    unsigned cur[DEPTH][2], nxt[DEPTH][2];
    float cur_alpha[DEPTH], nxt_alpha[DEPTH];
    // A pair's 6 bytes are fetched by one 8-byte load (2-byte aligned; the 2
    // bytes beyond belong to the next pair and are ignored) -- except in the
    // wave that holds the last pair of the picture, which would read past
    // the end of the last frame: that one takes three 2-byte loads.
    const bool last_wave =
        __builtin_amdgcn_ballot_w64(i0 + 2 >= n_pixels) != 0;
    // The loads go through a buffer descriptor per group of frames: the
    // group's base lives in scalar registers, the frame within the group is
    // a scalar offset and the lane's pair a 32-bit vector offset that never
    // changes, so a load costs no vector address arithmetic (a 64-bit vector
    // multiply-add per frame before).  The host checks that DEPTH frames
    // span less than 4 GB.
    // Odd lanes load from 2 bytes before their pair: every load is 4-byte
    // aligned when the frames are, and none reaches beyond the wave's own row
    // segments into a cache line of the neighbouring group.  decode() shifts
    // the two surplus bytes out.
    const unsigned sh = last_wave ? 0u : (lane & 1) * 16u;
    const int src_off = (int)(i0 * 3) - (int)(sh >> 3);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto fetch_group = [&](int t, unsigned (&d)[DEPTH][2]) {
        // frames t .. t + DEPTH - 1; past the end of the batch the last
        // whole group is read again (nobody looks at the result)
        if (t + DEPTH > nframes) t = nframes >= DEPTH ? nframes - DEPTH : 0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(bgr + (long)t * bgr_frame_stride), 0,
            HK_MOG_FRAME_RECORDS, 0x00020000);
        // (the host launches this kernel with nframes >= DEPTH)
        if (!last_wave) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const u2 v = __builtin_bit_cast(
                    u2, __builtin_amdgcn_raw_buffer_load_b64(
                            rsrc, src_off, (int)(j * bgr_frame_stride), 0));
                d[j][0] = v.x;
                d[j][1] = v.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int so = (int)(j * bgr_frame_stride);
                const unsigned a = __builtin_amdgcn_raw_buffer_load_b16(
                    rsrc, src_off, so, 0);
                const unsigned b = __builtin_amdgcn_raw_buffer_load_b16(
                    rsrc, src_off + 2, so, 0);
                const unsigned e = __builtin_amdgcn_raw_buffer_load_b16(
                    rsrc, src_off + 4, so, 0);
                d[j][0] = a | (b << 16);
                d[j][1] = e;
            }
        }
    };
    auto alphas_of = [&](int t, float (&a)[DEPTH]) {
        int i = frames_before + t;          // frame t has n = i + 1
        i = i < HISTORY ? i : HISTORY;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) a[j] = alpha_table[i + j];
    };
    fetch_group(0, cur);
    alphas_of(0, cur_alpha);
    for (int t0 = 0; t0 < nframes; t0 += DEPTH) {
        fetch_group(t0 + DEPTH, nxt);
        alphas_of(t0 + DEPTH, nxt_alpha);
        // frame j of the group as three pairs of floats
        auto decode = [&](int j, f2 &x0, f2 &x1, f2 &x2) {
            const unsigned c01 =
                __builtin_amdgcn_alignbit(cur[j][1], cur[j][0], sh);
            const unsigned c2 = cur[j][1] >> sh;
            // bytes b0 g0 r0 b1 | g1 r1 (and two of the next pair)
            x0.x = (float)(c01 & 255u);
            x1.x = (float)((c01 >> 8) & 255u);
            x2.x = (float)((c01 >> 16) & 255u);
            x0.y = (float)(c01 >> 24);
            x1.y = (float)(c2 & 255u);
            x2.y = (float)((c2 >> 8) & 255u);
        };
        // The frames of the group a wave in unit mode gets through with every
        // lane near its component: straight-line code on the mean and the
        // variance alone, up to the first frame that does not qualify.
        int unit_frames = 0;
        if (mode <= 0) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                if (unit_frames == j) {
                    f2 x0, x1, x2;
                    decode(j, x0, x1, x2);
                    const Probe pr = probe0(px, x0, x1, x2);
                    const f2 thr = FAST_VT * vsum;
                    // (a ballot of a compare is the compare's own result;
                    // a ballot of their conjunction would cost two vector
                    // instructions more)
                    const unsigned long long near =
                        __builtin_amdgcn_ballot_w64(pr.dist2.x < thr.x) &
                        __builtin_amdgcn_ballot_w64(pr.dist2.y < thr.y);
                    if (near == all_lanes) {
                        asm volatile("; unit step");
                        unit_update(px, vsum, pr, cur_alpha[j]);
                        HK_PATH_STAT(0);
                        unit_frames = j + 1;
                        if (MASKS)      // all background
                            *reinterpret_cast<unsigned short *>(
                                dst + (long)(t0 + j) * mask_frame_stride) = 0;
                    }
                }
            }
            if (unit_frames) mode = -1;
        }
        if (unit_frames < DEPTH) {
            if (mode < 0) {         // see `mode`
                lazy.w = lazy.scale = f2{1.f, 1.f};
                mode = 0;
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                if (j < unit_frames) continue;
                const int t = t0 + j;
                const float alpha = cur_alpha[j];
                f2 x0, x1, x2;
                decode(j, x0, x1, x2);
                const Probe pr = probe0(px, x0, x1, x2);
                const f2 thr = FAST_VT * vsum;
                // weights are never negative: their bit patterns order like
                // the values
                const unsigned wmin = min(__float_as_uint(px.f[F_W][0].x),
                                          __float_as_uint(px.f[F_W][0].y));
                const bool hit = pr.dist2.x < thr.x && pr.dist2.y < thr.y &&
                                 wmin >= __float_as_uint(FLT_EPSILON);
                const unsigned long long hits =
                    __builtin_amdgcn_ballot_w64(pr.dist2.x < thr.x) &
                    __builtin_amdgcn_ballot_w64(pr.dist2.y < thr.y) &
                    __builtin_amdgcn_ballot_w64(
                        wmin >= __float_as_uint(FLT_EPSILON));
                const int nc = mode < 1 ? 1 : mode;
                // cum > T = background.  The fast step leaves each pixel's
                // cumulated weights there, the general step a stand-in on
                // the right side of T, so that one compare per pixel after
                // the paths have joined yields the wave's foreground bits.
                f2 cum;
                bool recount = false;
                if (hits == all_lanes) {
                    asm volatile("; every lane on its first component");
                    cum = weighted_update(px, lazy, vsum, nc, pr, alpha);
                    HK_PATH_STAT(nc < 3 ? nc : 3);
                } else {
                    if (hit) {
                        cum = weighted_update(px, lazy, vsum, nc, pr, alpha);
                    } else {
                        // the launch ends when its slowest wave does: waves
                        // in the long general step go first at instruction
                        // issue
                        __builtin_amdgcn_s_setprio(3);
                        materialise_sk0(px, lazy, vsum);
                        Lane<0> l0{px};
                        Lane<1> l1{px};
                        cum.x = mog_step(l0, x0.x, x1.x, x2.x, alpha, c)
                                    ? 0.f : 1.f;
                        cum.y = mog_step(l1, x0.y, x1.y, x2.y, alpha, c)
                                    ? 0.f : 1.f;
                        vsum = (px.f[F_V0][0] + px.f[F_V1][0]) +
                               px.f[F_V2][0];
                        __builtin_amdgcn_s_setprio(0);
                    }
                    ++slow_steps;
                    recount = true;
                    HK_PATH_STAT(__builtin_amdgcn_ballot_w64(hit) ==
                                         all_lanes ? 4 : 5);
                }
                // components only ever get more, and only in the general
                // step; a wave of single components is looked at after every
                // frame until its weights are all 1
                if (recount || mode == 1) mode = wave_mode();
                const bool f0 = !(cum.x > FAST_T), f1 = !(cum.y > FAST_T);
                if (MASKS) {
                    const unsigned m = (f0 ? 255u : 0u) |
                                       (f1 ? 255u << 8 : 0u);
                    *reinterpret_cast<unsigned short *>(
                        dst + (long)t * mask_frame_stride) = (unsigned short)m;
                }
                if (TILES) {
                    const unsigned long long b0 =
                        __builtin_amdgcn_ballot_w64(f0);
                    const unsigned long long b1 =
                        __builtin_amdgcn_ballot_w64(f1);
                    if ((b0 | b1) != 0ull)
                        write_lane_x4(ta0, ta1, ta2, ta3, t & 63, (unsigned)b0,
                                      (unsigned)(b0 >> 32), (unsigned)b1,
                                      (unsigned)(b1 >> 32));
                }
            }
        }
        if (TILES) {
            // frames [first, done) of the current group of 64 are in the
            // accumulators; store when the group is complete or the batch ends
            const int done = t0 + DEPTH < nframes ? t0 + DEPTH : nframes;
            if ((done & 63) == 0 || done == nframes) {
                const int first = (done - 1) & ~63;
                if (first + lane < done)
                    *reinterpret_cast<uint4 *>(
                        tile_dst + (long)first * tile_words_per_frame) =
                        make_uint4(ta0, ta1, ta2, ta3);
                ta0 = ta1 = ta2 = ta3 = 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            cur[j][0] = nxt[j][0];
            cur[j][1] = nxt[j][1];
            cur_alpha[j] = nxt_alpha[j];
        }
    }

    if (cost && threadIdx.x == 0 && slow_steps)
        atomicMax(&cost[group], slow_steps);
#ifdef HK_MOG_WAVE_TIMES
    if (hk_mog_wave_times_dev && DEPTH > 1) {
        unsigned long long wave_t1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(wave_t1)::"memory");
        if (threadIdx.x == 0) {
            unsigned long long *rec =
                hk_mog_wave_times_dev + 4 * (unsigned long long)blockIdx.x;
            rec[0] = wave_t0;
            rec[1] = wave_t1;
            rec[2] = (unsigned long long)slow_steps |
                     ((unsigned long long)(unsigned)mode << 32);
            rec[3] = pos;
        }
    }
#endif
    if (mode < 0) lazy.w = lazy.scale = f2{1.f, 1.f};
    materialise_sk0(px, lazy, vsum);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        *reinterpret_cast<float2 *>(base + (long)(F_W * K + k) * n_pixels) =
            make_float2(px.f[F_W][k].x, px.f[F_W][k].y);
        const bool used = k == 0 || px.f[F_W][k].x != 0.f ||
                          px.f[F_W][k].y != 0.f;
        if (used) {
#pragma unroll
            for (int q = F_SK; q < 8; ++q) {
                float2 v;
                if (is_cold(q, k)) {
                    v.x = px.cold[cold_slot(q, k)];
                    v.y = px.cold[cold_slot(q, k) + 1];
                } else {
                    v.x = px.f[q][k].x;
                    v.y = px.f[q][k].y;
                }
                *reinterpret_cast<float2 *>(
                    base + (long)(q * K + k) * n_pixels) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Launch order.  A wave that keeps leaving the fast step (its tile lies on a
// moving object or on its edge) runs several times longer than the others;
// there are only ~4 waves per slot in a launch, so where such waves start
// decides when the launch ends.  Objects move little from one batch of frames
// to the next: each launch records per block the largest number of slow
// frames among its waves, and the next launch starts the blocks in 9 classes
// of decreasing cost (the workgroup dispatcher hands blocks out in index
// order).  Scheduling only: results do not depend on it.
// One workgroup; reads cost[], writes order[], zeroes cost[].
// ---------------------------------------------------------------------------
constexpr int ORDER_CLASSES = 9;
constexpr int ORDER_LDS_GROUPS = 48 * 1024;

__global__ __launch_bounds__(1024) void mog_order_k(int *__restrict__ cost,
                                                    int *__restrict__ order,
                                                    int n_blocks,
                                                    int frames_of_cost) {
    __shared__ int count[ORDER_CLASSES], start[ORDER_CLASSES];
    // Each group's class is decided ONCE (one read of cost[b]) and kept here
    // for the placement pass: order[] is then a permutation whatever happens
    // to cost[] meanwhile (hk_mog_apply_batch_static shares one cost[] among
    // all callers of a GPU; a call on another stream may write it).  Groups
    // beyond the array (frames of more than 25 M pixels) keep launch order.
    __shared__ unsigned char cls_of[ORDER_LDS_GROUPS];
    if (threadIdx.x < ORDER_CLASSES) count[threadIdx.x] = 0;
    __syncthreads();
    auto cls = [&](int c) {
        if (c <= 0) return ORDER_CLASSES - 1;
        int q = (int)((long)c * (ORDER_CLASSES - 1) / (frames_of_cost + 1));
        q = q < ORDER_CLASSES - 2 ? q : ORDER_CLASSES - 2;
        return ORDER_CLASSES - 2 - q;           // 0 = slowest
    };
    const int n_sorted = n_blocks < ORDER_LDS_GROUPS ? n_blocks
                                                     : ORDER_LDS_GROUPS;
    for (int b = threadIdx.x; b < n_sorted; b += blockDim.x) {
        const int k = cls(cost[b]);
        cls_of[b] = (unsigned char)k;
        atomicAdd(&count[k], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < ORDER_CLASSES; ++i) {
            start[i] = acc;
            acc += count[i];
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
        const int pos = b < n_sorted ? atomicAdd(&start[cls_of[b]], 1) : b;
        order[pos] = b;
        cost[b] = 0;
    }
}

// ---------------------------------------------------------------------------
// General path: any pitch / alignment / size; one pixel per thread.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mog_batch1_k(
    float *__restrict__ planes, long n_pixels, int width,
    const uint8_t *__restrict__ bgr, long bgr_frame_stride, long bgr_pitch,
    uint8_t *__restrict__ mask, long mask_frame_stride, long mask_pitch,
    int nframes, int frames_before, const float *__restrict__ alpha_table,
    const MogConst c) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const long y = i / width;
    const long x = i - y * width;

    Mix px;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k)
            px.f[q][k] = planes[(long)(q * K + k) * n_pixels + i];

    const uint8_t *src = bgr + y * bgr_pitch + x * 3;
    uint8_t *dst = mask + y * mask_pitch + x;
    for (int t = 0; t < nframes; ++t) {
        const uint8_t *s = src + (long)t * bgr_frame_stride;
        int n = frames_before + t + 1;
        n = n < HISTORY ? n : HISTORY;
        const unsigned m = mog_step(px, (float)s[0], (float)s[1], (float)s[2],
                                    alpha_table[n - 1], c);
        dst[(long)t * mask_frame_stride] = (uint8_t)m;
    }

#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k)
            planes[(long)(q * K + k) * n_pixels + i] = px.f[q][k];
}

// ---------------------------------------------------------------------------
// Self-check of the trimmed sqrt / divide sequences against the IEEE
// operators (which this file is compiled to round correctly) over the ranges
// the packed path uses them on, and wider.
//   part 0: sqrt_ranged(x)      for every float in [2, 2^22)
//   part 1: div_ranged(1, x)    for every float in [2^-26, 2^6)
//   part 2: div_ranged(n, d)    n in [2^-26, 4), d in [1, 2^12), hashed samples
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void mog_math_check_k(
    int part, unsigned lo_bits, unsigned long long count,
    unsigned long long *mismatches) {
    unsigned long long bad = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x +
                                threadIdx.x; i < count; i += stride) {
        if (part == 0) {
            const float x = __uint_as_float(lo_bits + (unsigned)i);
            bad += __float_as_uint(sqrt_ranged(x)) != __float_as_uint(sqrtf(x));
        } else if (part == 1) {
            const float x = __uint_as_float(lo_bits + (unsigned)i);
            bad += __float_as_uint(div_ranged(1.f, x)) !=
                   __float_as_uint(1.f / x);
        } else {
            const unsigned h0 = mix32((unsigned)i * 2u + 1u);
            const unsigned h1 = mix32(h0 ^ (unsigned)(i >> 32) ^ 0x9e3779b9u);
            // n: exponent 2^-26 .. 2^1, d: exponent 2^0 .. 2^11
            const unsigned ne = 101u + (h0 >> 23) % 28u;
            const unsigned de = 127u + (h1 >> 23) % 12u;
            const float n = __uint_as_float((ne << 23) | (h0 & 0x7fffffu));
            const float d = __uint_as_float((de << 23) | (h1 & 0x7fffffu));
            bad += __float_as_uint(div_ranged(n, d)) != __float_as_uint(n / d);
            f2 n2 = {n, d}, d2 = {d, n + 1.f};
            const f2 q = div_ranged2(n2, d2);
            bad += __float_as_uint(q.x) != __float_as_uint(n / d);
            bad += __float_as_uint(q.y) != __float_as_uint(d / (n + 1.f));
            f2 s2 = {d, d * 3.f};
            const f2 r = sqrt_ranged2(s2);
            bad += __float_as_uint(r.x) != __float_as_uint(sqrtf(d));
            bad += __float_as_uint(r.y) != __float_as_uint(sqrtf(d * 3.f));
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// constants exactly as bgfg_gaussmix.cpp derives them
static MogConst opencv_defaults() {
    const double noiseSigma = 30 * 0.5;
    MogConst c;
    c.T = (float)0.7;
    c.vT = (float)(2.5 * 2.5);
    c.w0 = (float)0.05;
    c.sk0 = (float)(c.w0 / (noiseSigma * 2 * std::sqrt(3.)));
    c.var0 = (float)(noiseSigma * noiseSigma * 4);
    c.minVar = (float)(noiseSigma * noiseSigma);
    return c;
}

struct hk_mog {
    int width = 0, height = 0, device = 0;
    long n_pixels = 0;
    int nframes = 0;
    float *planes = nullptr;        // [NPLANES][n_pixels]
    float *alpha_table = nullptr;   // [ALPHA_ENTRIES]
    int *cost = nullptr;            // [n_groups] slow frames, last batch launch
    int *order = nullptr;           // [n_groups] launch order, see mog_order_k
    int n_groups = 0;
    int frames_of_cost = 0;         // frames of the launch that wrote cost[]
    MogConst c;
};

extern "C" int hk_mog_create(int width, int height, int device, hk_mog **out) {
    if (!out || width <= 0 || height <= 0)
        return hk::fail(HK_ERR_ARG, "hk_mog_create: bad size %dx%d", width,
                        height);
    hk::DeviceGuard guard(device);
    if (!guard.ok)
        return hk::fail(HK_ERR_HIP, "hk_mog_create: cannot select device %d",
                        device);
    hk_mog *m = new hk_mog();
    m->width = width;
    m->height = height;
    m->device = device;
    m->n_pixels = (long)width * height;
    m->c = opencv_defaults();
    if (m->c.T != FAST_T || m->c.vT != FAST_VT || m->c.minVar != FAST_MINVAR) {
        delete m;
        return hk::fail(HK_ERR_INTERNAL, "hk_mog_create: constants");
    }

    const size_t bytes = sizeof(float) * NPLANES * (size_t)m->n_pixels;
    hipError_t e = hipMalloc((void **)&m->planes, bytes);
    if (e == hipSuccess) e = hipMemset(m->planes, 0, bytes);
    if (e == hipSuccess)
        e = hipMalloc((void **)&m->alpha_table, sizeof(float) * ALPHA_ENTRIES);
    m->n_groups = (int)((m->n_pixels / 2 + 255) / 256);
    if (e == hipSuccess)
        e = hipMalloc((void **)&m->cost, sizeof(int) * 2 * m->n_groups);
    if (e == hipSuccess) {
        m->order = m->cost + m->n_groups;
        e = hipMemset(m->cost, 0, sizeof(int) * 2 * m->n_groups);
    }
    if (e == hipSuccess) {
        float tbl[ALPHA_ENTRIES];
        for (int i = 0; i < ALPHA_ENTRIES; ++i)
            tbl[i] = (float)(1. / (i < HISTORY ? i + 1 : HISTORY));
        e = hipMemcpy(m->alpha_table, tbl, sizeof(tbl), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(m->planes);
        (void)hipFree(m->alpha_table);
        (void)hipFree(m->cost);
        delete m;
        return hk::fail(HK_ERR_ALLOC, "hk_mog_create: %s",
                        hipGetErrorString(e));
    }
    *out = m;
    return HK_OK;
}

extern "C" int hk_mog_destroy(hk_mog *m) {
    if (!m) return HK_OK;
    hk::DeviceGuard guard(m->device);
    (void)hipFree(m->planes);
    (void)hipFree(m->alpha_table);
    (void)hipFree(m->cost);
    delete m;
    return HK_OK;
}

extern "C" int hk_mog_reset(hk_mog *m, void *stream) {
    if (!m) return hk::fail(HK_ERR_ARG, "hk_mog_reset: null handle");
    hk::DeviceGuard guard(m->device);
    HK_HIP(hipMemsetAsync(m->planes, 0,
                          sizeof(float) * NPLANES * (size_t)m->n_pixels,
                          (hipStream_t)stream));
    HK_HIP(hipMemsetAsync(m->cost, 0, sizeof(int) * m->n_groups,
                          (hipStream_t)stream));
    m->nframes = 0;
    m->frames_of_cost = 0;
    return HK_OK;
}

extern "C" int hk_mog_nframes(const hk_mog *m) { return m ? m->nframes : 0; }

// What one batch launch works on: the model, the alpha table and the launch
// order bookkeeping (cost / order, see mog_order_k).  hk_mog keeps these in
// buffers of its own; hk_mog_apply_batch_static takes the model from the
// caller and the rest from the library's static device array.
struct MogTarget {
    int width, height;
    long n_pixels;
    float *planes;
    const float *alpha_table;
    int *cost, *order;          // may be null: no launch order kept
    int n_groups;
    int *frames_of_cost;        // host: frames of the launch that wrote cost[]
    int frames_before;
    const MogConst *c;
};

// mask_dev and tiles_dev: either may be null, not both (tiles only with the
// tiled kernel).
static int launch_batch(const MogTarget &m, const void *bgr_dev,
                        long bgr_frame_stride, long bgr_pitch, void *mask_dev,
                        long mask_frame_stride, long mask_pitch,
                        void *tiles_dev, int nframes, hipStream_t s) {
    if (!mask_dev) {            // tiles only: no mask geometry to check
        mask_pitch = m.width;
        mask_frame_stride = (long)m.width * m.height;
    }
    if (nframes < 0)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch: nframes %d", nframes);
    if (nframes == 0) return HK_OK;
    if (bgr_pitch < 3L * m.width || mask_pitch < m.width)
        return hk::fail(HK_ERR_ARG,
                        "hk_mog_apply_batch: pitch too small (bgr %ld, mask "
                        "%ld) for width %d", bgr_pitch, mask_pitch, m.width);
    if (nframes > 1 && (bgr_frame_stride < bgr_pitch * m.height ||
                        mask_frame_stride < mask_pitch * m.height))
        return hk::fail(HK_ERR_ARG,
                        "hk_mog_apply_batch: frame stride smaller than a "
                        "frame");
    const long N = m.n_pixels;
    int frames_before = m.frames_before;
    const bool fast =
        bgr_pitch == 3L * m.width && mask_pitch == m.width && N % 2 == 0 &&
        3 * N < (1L << 31) && (nframes == 1 || bgr_frame_stride < (1L << 29)) &&
        ((uintptr_t)bgr_dev & 1) == 0 && ((uintptr_t)mask_dev & 1) == 0 &&
        (nframes == 1 ||
         (bgr_frame_stride % 2 == 0 && mask_frame_stride % 2 == 0));
    if (fast) {
        const unsigned blocks = (unsigned)((m.n_groups + 7) / 8) * 8 * 4;
        // short launches stream the model (HBM-bound), long ones are
        // bound by the latency of the frame loads and by arithmetic: see
        // mog_batch2_k
#define HK_MOG_LAUNCH(W, D, TL, MK, TS)                                       \
        hipLaunchKernelGGL((mog_batch2_k<W, D, TL, MK, TS>), dim3(blocks),    \
                           dim3(64), 0, s, m.planes, N, m.width,              \
                           (const uint8_t *)bgr_dev,                          \
                           bgr_frame_stride, (uint8_t *)mask_dev,             \
                           mask_frame_stride, nframes, frames_before,         \
                           m.alpha_table, *m.c, m.n_groups, order, cost,      \
                           (unsigned *)tiles_dev, tile_words)
        // (HK_MOG_UNTILED: measurements only -- 128 x 1 pixel units, whose
        // rows are whole cache lines, on a geometry that has 32 x 4 tiles)
        static const bool force_untiled = getenv("HK_MOG_UNTILED") != nullptr;
        const bool tileable = m.width % 32 == 0 && m.height % 4 == 0 &&
                              !(force_untiled && !tiles_dev);
        const long tile_words = (N / 128) * 4;
        if (tiles_dev && !tileable)
            return hk::fail(HK_ERR_ARG,
                            "hk_mog_apply_batch_tiles: %dx%d frames are not "
                            "made of 32x4 tiles", m.width, m.height);
        // batches are launched slowest groups first (cost of the previous
        // batch); single frames are not worth the bookkeeping
        const int *order = nullptr;
        int *cost = nullptr;
        if (nframes > 4 && m.cost) {
            cost = m.cost;
            if (*m.frames_of_cost > 0) {
                hipLaunchKernelGGL(mog_order_k, dim3(1), dim3(1024), 0, s,
                                   m.cost, m.order, m.n_groups,
                                   *m.frames_of_cost);
                HK_LAUNCH_CHECK("mog_order_k");
                order = m.order;
            }
            *m.frames_of_cost = nframes;
        }
        // the 4-frame-group kernel takes whole groups; what is left over
        // (and short batches) goes through the frame-at-a-time kernel
        const int n4 = nframes > 4 ? nframes & ~3 : 0;
        if (n4) {
            const int nframes_all = nframes;
            nframes = n4;
            if (tiles_dev) HK_MOG_LAUNCH(4, 4, true, false, true);
            else if (tileable) HK_MOG_LAUNCH(4, 4, true, true, false);
            else HK_MOG_LAUNCH(4, 4, false, true, false);
            HK_LAUNCH_CHECK("mog_batch2_k");
            nframes = nframes_all - n4;
            frames_before += n4;
            bgr_dev = (const uint8_t *)bgr_dev + (long)n4 * bgr_frame_stride;
            if (mask_dev)
                mask_dev = (uint8_t *)mask_dev + (long)n4 * mask_frame_stride;
            if (tiles_dev)
                tiles_dev = (unsigned *)tiles_dev + (long)n4 * tile_words;
            order = nullptr;        // scheduling only: not worth it here,
            cost = nullptr;         // and the costs are those of the groups
        }
        if (nframes > 0) {
            if (tiles_dev) HK_MOG_LAUNCH(4, 1, true, false, true);
            else if (tileable) HK_MOG_LAUNCH(4, 1, true, true, false);
            else HK_MOG_LAUNCH(4, 1, false, true, false);
        }
#undef HK_MOG_LAUNCH
        HK_LAUNCH_CHECK("mog_batch2_k");
    } else {
        if (tiles_dev)
            return hk::fail(HK_ERR_ARG,
                            "hk_mog_apply_batch_tiles: needs densely packed, "
                            "2-byte aligned frames less than 512 MB apart");
        const unsigned blocks = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(mog_batch1_k, dim3(blocks), dim3(256), 0, s,
                           m.planes, N, m.width, (const uint8_t *)bgr_dev,
                           bgr_frame_stride, bgr_pitch, (uint8_t *)mask_dev,
                           mask_frame_stride, mask_pitch, nframes,
                           m.frames_before, m.alpha_table, *m.c);
        HK_LAUNCH_CHECK("mog_batch1_k");
    }
    return HK_OK;
}

static int apply_batch(hk_mog *m, const void *bgr_dev, long bgr_frame_stride,
                       long bgr_pitch, void *mask_dev, long mask_frame_stride,
                       long mask_pitch, void *tiles_dev, int nframes,
                       void *stream) {
    if (!m || !bgr_dev || (!mask_dev && !tiles_dev))
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch: null argument");
    hk::DeviceGuard guard(m->device);
    const MogTarget t{m->width, m->height, m->n_pixels, m->planes,
                      m->alpha_table, m->cost, m->order, m->n_groups,
                      &m->frames_of_cost, m->nframes, &m->c};
    const int rc = launch_batch(t, bgr_dev, bgr_frame_stride, bgr_pitch,
                                mask_dev, mask_frame_stride, mask_pitch,
                                tiles_dev, nframes, (hipStream_t)stream);
    if (rc != HK_OK) return rc;
    if (nframes > 0) m->nframes += nframes;
    return HK_OK;
}

extern "C" int hk_mog_apply_batch(hk_mog *m, const void *bgr_dev,
                                  long bgr_frame_stride, long bgr_pitch,
                                  void *mask_dev, long mask_frame_stride,
                                  long mask_pitch, int nframes, void *stream) {
    if (!mask_dev)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch: null argument");
    return apply_batch(m, bgr_dev, bgr_frame_stride, bgr_pitch, mask_dev,
                       mask_frame_stride, mask_pitch, nullptr, nframes, stream);
}

extern "C" int hk_mog_tiles_supported(const hk_mog *m) {
    return m && m->width % 32 == 0 && m->height % 4 == 0 ? 1 : 0;
}

extern "C" int hk_mog_apply_batch_tiles(hk_mog *m, const void *bgr_dev,
                                        long bgr_frame_stride, long bgr_pitch,
                                        void *tiles_dev, int nframes,
                                        void *stream) {
    if (!tiles_dev)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch_tiles: null argument");
    if (m && ((uintptr_t)tiles_dev & 15))
        return hk::fail(HK_ERR_ARG,
                        "hk_mog_apply_batch_tiles: tiles must be 16-byte "
                        "aligned");
    return apply_batch(m, bgr_dev, bgr_frame_stride, bgr_pitch, nullptr, 0, 0,
                       tiles_dev, nframes, stream);
}

// ---------------------------------------------------------------------------
// hk_mog_apply_batch_static: the batch launch for callers that cannot call
// hipMalloc (code under HIP graph capture, tight serving loops, harnesses that
// forbid it).  The caller owns the model; everything else the launch needs
// beyond inputs and outputs -- the alpha table and the launch-order
// bookkeeping cost[] / order[] -- lives in `scratch`, a static device array
// of this code object.  Every GPU the library runs on holds its own copy of
// the array at its own address until the process ends.
//
// In host code the array's NAME is only a stand-in (passing it to a kernel
// would compile and then fault on the GPU): its device address is looked up
// once per device with hipGetSymbolAddress and only that pointer is used.
// ---------------------------------------------------------------------------
namespace {

constexpr long STATIC_MAX_PIXELS = 3840L * 2160L;       // one 4K frame
constexpr int STATIC_MAX_GROUPS = (int)((STATIC_MAX_PIXELS / 2 + 255) / 256);
constexpr size_t STATIC_ALPHA_BYTES = 1024;             // HISTORY floats, padded
constexpr size_t STATIC_SCRATCH_BYTES =
    STATIC_ALPHA_BYTES + 2 * sizeof(int) * (size_t)STATIC_MAX_GROUPS;   // 128 KB
static_assert(sizeof(float) * ALPHA_ENTRIES <= STATIC_ALPHA_BYTES,
              "alpha table");

}  // namespace

static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {

// alpha[i] = (float)(1. / min(i + 1, HISTORY)), as hk_mog_create computes it on
// the host
// (IEEE double division, then one rounding to float).
__global__ __launch_bounds__(256) void mog_alpha_table_k(float *alpha) {
    const int i = threadIdx.x;
    if (i < ALPHA_ENTRIES)
        alpha[i] = (float)(1. / (double)(i < HISTORY ? i + 1 : HISTORY));
}

constexpr int STATIC_MAX_DEVICES = 64;

struct StaticState {            // per device, guarded by static_mutex
    unsigned char *ptr = nullptr;       // device address of `scratch` there
    const void *model = nullptr;        // model of the call that wrote cost[]
    int n_groups = 0;
    int frames_of_cost = 0;
    int frames_after = 0;       // frames the model had seen after that call
};
StaticState static_state[STATIC_MAX_DEVICES];
std::mutex static_mutex;

}  // namespace

extern "C" int hk_mog_apply_batch_static(
    int width, int height, int device, void *model_dev, int frames_before,
    const void *bgr_dev, long bgr_frame_stride, long bgr_pitch, void *mask_dev,
    long mask_frame_stride, long mask_pitch, int nframes, void *stream) {
    if (width <= 0 || height <= 0)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch_static: bad size %dx%d",
                        width, height);
    if (!model_dev || !bgr_dev || !mask_dev)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch_static: null argument");
    if (frames_before < 0)
        return hk::fail(HK_ERR_ARG,
                        "hk_mog_apply_batch_static: frames_before %d",
                        frames_before);
    if ((uintptr_t)model_dev & 7)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch_static: model must be "
                                    "8-byte aligned");
    if (device < 0 || device >= STATIC_MAX_DEVICES)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch_static: device %d "
                                    "outside 0..%d", device,
                        STATIC_MAX_DEVICES - 1);
    const long N = (long)width * height;
    if (N > STATIC_MAX_PIXELS)
        return hk::fail(HK_ERR_CAPACITY,
                        "hk_mog_apply_batch_static: %dx%d = %ld pixels exceed "
                        "the static scratch buffer's maximum of %ld (3840x2160); "
                        "use hk_mog_create + hk_mog_apply_batch", width, height,
                        N, STATIC_MAX_PIXELS);
    hk::DeviceGuard guard(device);
    if (!guard.ok)
        return hk::fail(HK_ERR_HIP, "hk_mog_apply_batch_static: cannot select "
                                    "device %d", device);
    std::lock_guard<std::mutex> lock(static_mutex);
    StaticState &st = static_state[device];
    if (!st.ptr) {              // once per device, with that device current
        void *p = nullptr;
        HK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p)
            return hk::fail(HK_ERR_HIP, "hk_mog_apply_batch_static: no device "
                                        "address for the scratch array");
        st.ptr = (unsigned char *)p;
    }
    const hipStream_t s = (hipStream_t)stream;
    float *const alpha = (float *)st.ptr;
    int *const cost = (int *)(st.ptr + STATIC_ALPHA_BYTES);
    int *const order = cost + STATIC_MAX_GROUPS;
    const int n_groups = (int)((N / 2 + 255) / 256);

    // The array keeps the last call's contents.  The alpha table is the same
    // for every call but is rewritten on this call's stream, so that a call
    // never reads what a call on another stream may still be writing.  cost[]
    // is reused (to order this launch, slowest groups first) only if the last
    // call that wrote it worked on the same model, the same geometry and the
    // frames right before these; otherwise it starts at zero.
    hipLaunchKernelGGL(mog_alpha_table_k, dim3(1), dim3(256), 0, s, alpha);
    HK_LAUNCH_CHECK("mog_alpha_table_k");
    static const MogConst consts = opencv_defaults();
    int frames_of_cost = 0;
    if (nframes > 4) {
        const bool continues = st.model == model_dev &&
                               st.n_groups == n_groups &&
                               st.frames_of_cost > 0 && frames_before > 0 &&
                               st.frames_after == frames_before;
        if (continues) frames_of_cost = st.frames_of_cost;
        else HK_HIP(hipMemsetAsync(cost, 0, sizeof(int) * n_groups, s));
    }
    const MogTarget t{width, height, N, (float *)model_dev, alpha, cost, order,
                      n_groups, &frames_of_cost, frames_before, &consts};
    const int rc = launch_batch(t, bgr_dev, bgr_frame_stride, bgr_pitch,
                                mask_dev, mask_frame_stride, mask_pitch,
                                nullptr, nframes, s);
    if (rc != HK_OK) return rc;
    if (nframes > 4) {
        st.model = model_dev;
        st.n_groups = n_groups;
        st.frames_of_cost = frames_of_cost;     // = nframes (fast path) or 0
        st.frames_after = frames_before + nframes;
    }
    return HK_OK;
}

extern "C" int hk_mog_apply(hk_mog *m, const void *bgr_dev, long bgr_pitch,
                            void *mask_dev, long mask_pitch, void *stream) {
    return hk_mog_apply_batch(m, bgr_dev, 0, bgr_pitch, mask_dev, 0,
                              mask_pitch, 1, stream);
}

extern "C" int hk_mog_export_model(hk_mog *m, float *host_out, void *stream) {
    if (!m || !host_out)
        return hk::fail(HK_ERR_ARG, "hk_mog_export_model: null argument");
    hk::DeviceGuard guard(m->device);
    const long N = m->n_pixels;
    std::vector<float> soa((size_t)NPLANES * N);
    HK_HIP(hipMemcpyAsync(soa.data(), m->planes, sizeof(float) * soa.size(),
                          hipMemcpyDeviceToHost, (hipStream_t)stream));
    HK_HIP(hipStreamSynchronize((hipStream_t)stream));
    // OpenCV's MixData order: sortKey, weight, mean[3], var[3]
    static const int aos_field[8] = { F_SK, F_W, F_M0, F_M1, F_M2,
                                      F_V0, F_V1, F_V2 };
    for (long i = 0; i < N; ++i)
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < 8; ++j)
                host_out[(i * K + k) * 8 + j] =
                    soa[(size_t)(aos_field[j] * K + k) * N + i];
    return HK_OK;
}

extern "C" int hk_mog_selfcheck_math(int device, long long *mismatches_out) {
    if (!mismatches_out)
        return hk::fail(HK_ERR_ARG, "hk_mog_selfcheck_math: null argument");
    hk::DeviceGuard guard(device);
    if (!guard.ok)
        return hk::fail(HK_ERR_HIP,
                        "hk_mog_selfcheck_math: cannot select device %d",
                        device);
    unsigned long long *d_bad = nullptr;
    HK_HIP(hipMalloc((void **)&d_bad, sizeof(*d_bad)));
    hipError_t e = hipMemset(d_bad, 0, sizeof(*d_bad));
    auto bits = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
    const struct { int part; unsigned lo; unsigned long long n; } runs[3] = {
        {0, bits(2.f), (unsigned long long)bits(4194304.f) - bits(2.f)},
        {1, bits(1.4901161e-8f), (unsigned long long)bits(64.f) -
                                     bits(1.4901161e-8f)},
        {2, 0u, 1ull << 30},
    };
    for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        hipLaunchKernelGGL(mog_math_check_k, dim3(256 * 32), dim3(256), 0, 0,
                           runs[i].part, runs[i].lo, runs[i].n, d_bad);
        e = hipGetLastError();
    }
    unsigned long long bad = 0;
    if (e == hipSuccess)
        e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess)
        return hk::fail(HK_ERR_HIP, "hk_mog_selfcheck_math: %s",
                        hipGetErrorString(e));
    *mismatches_out = (long long)bad;
    return HK_OK;
}


#ifdef HK_MOG_WAVE_TIMES
// records: [workgroups of the batch launch][4] = start, end, slow frames |
// mode << 32, position in the launch order; null switches the recording off
extern "C" int hk_mog_wave_times(unsigned long long *dev_records) {
    HK_HIP(hipDeviceSynchronize());
    HK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(hk_mog_wave_times_dev), &dev_records,
                             sizeof(dev_records)));
    return HK_OK;
}
#endif

#ifdef HK_MOG_WAVE_TIMES
extern "C" int hk_mog_frame_records(int records) {
    HK_HIP(hipDeviceSynchronize());
    HK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(hk_mog_frame_records_dev), &records,
                             sizeof(records)));
    return HK_OK;
}
#endif

#ifdef HK_MOG_PATH_STATS
extern "C" int hk_mog_path_stats(unsigned long long *out) {
    unsigned long long zero[8] = {0};
    HK_HIP(hipDeviceSynchronize());
    HK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(hk_mog_path_stats_dev),
                               sizeof(zero)));
    HK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(hk_mog_path_stats_dev), zero,
                             sizeof(zero)));
    return HK_OK;
}
#endif
