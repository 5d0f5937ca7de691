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
// The recursion is independent per pixel, so one launch can advance a pixel
// through many frames with its whole mixture held in registers: per frame
// only the 3 input bytes are read and the mask byte written.  hk_mog_apply is
// the same kernel with one frame.
//
// Arithmetic is bit-identical to the oracle: same fp32 operation order, no
// FMA contraction (-ffp-contract=off), IEEE divide and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved.

#include <cfloat>
#include <cstdint>
#include <vector>

#include "hk_common.h"

namespace {

constexpr int K = 5;
constexpr int NPLANES = 8 * K;
constexpr int HISTORY = 200;

enum Field { F_W = 0, F_SK, F_M0, F_M1, F_M2, F_V0, F_V1, F_V2 };

// OpenCV's defaults, computed exactly as bgfg_gaussmix.cpp computes them.
struct MogConst {
    float T, vT, w0, sk0, var0, minVar;
};

struct Mix {                // one pixel's mixture, lives in VGPRs
    float f[8][K];          // [field][component]; every index is a
                            // compile-time constant after unrolling
};

__device__ __forceinline__ float max_like_std(float a, float b) {
    return (a < b) ? b : a;             // std::max(a, b)
}

// One frame of one pixel.  Returns 255 (foreground) or 0.
__device__ __forceinline__ unsigned mog_step(Mix &p, float x0, float x1,
                                             float x2, float alpha,
                                             const MogConst c) {
    float wsum = 0.f;
    int kHit = -1;
    int kEnd = K;           // k at which the scan stopped
    bool scanning = true;

#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (scanning) {
            const float w = p.f[F_W][k];
            wsum += w;
            if (w < FLT_EPSILON) {
                scanning = false;
                kEnd = k;
            } else {
                const float d0 = x0 - p.f[F_M0][k];
                const float d1 = x1 - p.f[F_M1][k];
                const float d2 = x2 - p.f[F_M2][k];
                float v0 = p.f[F_V0][k], v1 = p.f[F_V1][k], v2 = p.f[F_V2][k];
                float dist2 = d0 * d0;
                dist2 = dist2 + d1 * d1;
                dist2 = dist2 + d2 * d2;
                if (dist2 < c.vT * (v0 + v1 + v2)) {
                    wsum -= w;
                    const float dw = alpha * (1.f - w);
                    p.f[F_W][k] = w + dw;
                    p.f[F_M0][k] = p.f[F_M0][k] + alpha * d0;
                    p.f[F_M1][k] = p.f[F_M1][k] + alpha * d1;
                    p.f[F_M2][k] = p.f[F_M2][k] + alpha * d2;
                    v0 = max_like_std(v0 + alpha * (d0 * d0 - v0), c.minVar);
                    v1 = max_like_std(v1 + alpha * (d1 * d1 - v1), c.minVar);
                    v2 = max_like_std(v2 + alpha * (d2 * d2 - v2), c.minVar);
                    p.f[F_V0][k] = v0;
                    p.f[F_V1][k] = v1;
                    p.f[F_V2][k] = v2;
                    p.f[F_SK][k] = w / sqrtf(v0 + v1 + v2);

                    // keep components ordered by sortKey (insertion step)
                    int pos = k;
                    bool moving = true;
#pragma unroll
                    for (int k1 = k - 1; k1 >= 0; --k1) {
                        if (moving) {
                            if (p.f[F_SK][k1] >= p.f[F_SK][k1 + 1]) {
                                moving = false;
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; ++q) {
                                    const float t = p.f[q][k1];
                                    p.f[q][k1] = p.f[q][k1 + 1];
                                    p.f[q][k1 + 1] = t;
                                }
                                pos = k1;
                            }
                        }
                    }
                    kHit = pos;
                    // upstream adds the tail from index k of the REORDERED
                    // array; reproduced as is
#pragma unroll
                    for (int kk = k; kk < K; ++kk) wsum += p.f[F_W][kk];
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
                wsum += c.w0 - p.f[F_W][kk];
                p.f[F_W][kk] = c.w0;
                p.f[F_M0][kk] = x0;
                p.f[F_M1][kk] = x1;
                p.f[F_M2][kk] = x2;
                p.f[F_V0][kk] = c.var0;
                p.f[F_V1][kk] = c.var0;
                p.f[F_V2][kk] = c.var0;
                p.f[F_SK][kk] = c.sk0;
            }
        }
    }

    const float wscale = 1.f / wsum;
    wsum = 0.f;
    int kFg = -1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        p.f[F_W][k] *= wscale;
        wsum += p.f[F_W][k];
        p.f[F_SK][k] *= wscale;
        if (wsum > c.T && kFg < 0) kFg = k + 1;
    }
    return (kHit >= kFg) ? 255u : 0u;
}

// ---------------------------------------------------------------------------
// Fast path: contiguous frames, N % 2 == 0.  Each lane owns 2 consecutive
// pixels: three 2-byte loads and one 2-byte store per frame, float2 plane
// accesses once per launch.
//
// The kernel is bound by VALU issue, not by HBM (the model never leaves the
// registers during a launch), so what matters is waves per SIMD: with several
// waves resident, one wave's scalar/branch instructions and divide/sqrt
// latencies hide behind another's arithmetic.  WAVES = 4 caps the kernel at
// 128 VGPRs (a few rarely-live values spill to scratch): best for long
// batches.  WAVES = 2 (no spills) is best when a launch is a single frame and
// the 320 B/pixel of model traffic dominate.
// ---------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void mog_batch2_k(
    float *__restrict__ planes, long n_pixels, const uint8_t *__restrict__ bgr,
    long bgr_frame_stride, uint8_t *__restrict__ mask, long mask_frame_stride,
    int nframes, int frames_before, const float *__restrict__ alpha_table,
    const MogConst c) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;    // pair
    const long i0 = g * 2;
    if (i0 >= n_pixels) return;

    Mix px[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float2 v = *reinterpret_cast<const float2 *>(
                planes + (long)(q * K + k) * n_pixels + i0);
            px[0].f[q][k] = v.x;
            px[1].f[q][k] = v.y;
        }
    }

    const uint8_t *src = bgr + i0 * 3;
    uint8_t *dst = mask + i0;
    const unsigned short *s16 = reinterpret_cast<const unsigned short *>(src);
    unsigned c0 = s16[0], c1 = s16[1], c2 = s16[2];
    for (int t = 0; t < nframes; ++t) {
        unsigned n0 = c0, n1 = c1, n2 = c2;
        if (t + 1 < nframes) {          // next frame's bytes, ahead of use
            const unsigned short *nx = reinterpret_cast<const unsigned short *>(
                src + (long)(t + 1) * bgr_frame_stride);
            n0 = nx[0];
            n1 = nx[1];
            n2 = nx[2];
        }
        int n = frames_before + t + 1;
        n = n < HISTORY ? n : HISTORY;
        const float alpha = alpha_table[n - 1];

        // bytes b0 g0 | r0 b1 | g1 r1
        unsigned m;
        m = mog_step(px[0], (float)(c0 & 255u), (float)(c0 >> 8),
                     (float)(c1 & 255u), alpha, c);
        m |= mog_step(px[1], (float)(c1 >> 8), (float)(c2 & 255u),
                      (float)(c2 >> 8), alpha, c) << 8;
        *reinterpret_cast<unsigned short *>(dst + (long)t * mask_frame_stride) =
            (unsigned short)m;
        c0 = n0;
        c1 = n1;
        c2 = n2;
    }

#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float2 v;
            v.x = px[0].f[q][k];
            v.y = px[1].f[q][k];
            *reinterpret_cast<float2 *>(planes + (long)(q * K + k) * n_pixels +
                                        i0) = v;
        }
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

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

struct hk_mog {
    int width = 0, height = 0, device = 0;
    long n_pixels = 0;
    int nframes = 0;
    float *planes = nullptr;        // [NPLANES][n_pixels]
    float *alpha_table = nullptr;   // [HISTORY]
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
    // constants exactly as bgfg_gaussmix.cpp derives them
    const double noiseSigma = 30 * 0.5;
    m->c.T = (float)0.7;
    m->c.vT = (float)(2.5 * 2.5);
    m->c.w0 = (float)0.05;
    m->c.sk0 = (float)(m->c.w0 / (noiseSigma * 2 * std::sqrt(3.)));
    m->c.var0 = (float)(noiseSigma * noiseSigma * 4);
    m->c.minVar = (float)(noiseSigma * noiseSigma);

    const size_t bytes = sizeof(float) * NPLANES * (size_t)m->n_pixels;
    hipError_t e = hipMalloc((void **)&m->planes, bytes);
    if (e == hipSuccess) e = hipMemset(m->planes, 0, bytes);
    if (e == hipSuccess)
        e = hipMalloc((void **)&m->alpha_table, sizeof(float) * HISTORY);
    if (e == hipSuccess) {
        float tbl[HISTORY];
        for (int i = 0; i < HISTORY; ++i) tbl[i] = (float)(1. / (i + 1));
        e = hipMemcpy(m->alpha_table, tbl, sizeof(tbl), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(m->planes);
        (void)hipFree(m->alpha_table);
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
    delete m;
    return HK_OK;
}

extern "C" int hk_mog_reset(hk_mog *m, void *stream) {
    if (!m) return hk::fail(HK_ERR_ARG, "hk_mog_reset: null handle");
    hk::DeviceGuard guard(m->device);
    HK_HIP(hipMemsetAsync(m->planes, 0,
                          sizeof(float) * NPLANES * (size_t)m->n_pixels,
                          (hipStream_t)stream));
    m->nframes = 0;
    return HK_OK;
}

extern "C" int hk_mog_nframes(const hk_mog *m) { return m ? m->nframes : 0; }

extern "C" int hk_mog_apply_batch(hk_mog *m, const void *bgr_dev,
                                  long bgr_frame_stride, long bgr_pitch,
                                  void *mask_dev, long mask_frame_stride,
                                  long mask_pitch, int nframes, void *stream) {
    if (!m || !bgr_dev || !mask_dev)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch: null argument");
    if (nframes < 0)
        return hk::fail(HK_ERR_ARG, "hk_mog_apply_batch: nframes %d", nframes);
    if (nframes == 0) return HK_OK;
    if (bgr_pitch < 3L * m->width || mask_pitch < m->width)
        return hk::fail(HK_ERR_ARG,
                        "hk_mog_apply_batch: pitch too small (bgr %ld, mask "
                        "%ld) for width %d", bgr_pitch, mask_pitch, m->width);
    if (nframes > 1 && (bgr_frame_stride < bgr_pitch * m->height ||
                        mask_frame_stride < mask_pitch * m->height))
        return hk::fail(HK_ERR_ARG,
                        "hk_mog_apply_batch: frame stride smaller than a "
                        "frame");
    hk::DeviceGuard guard(m->device);
    const hipStream_t s = (hipStream_t)stream;
    const long N = m->n_pixels;
    const bool fast =
        bgr_pitch == 3L * m->width && mask_pitch == m->width && N % 2 == 0 &&
        ((uintptr_t)bgr_dev & 1) == 0 && ((uintptr_t)mask_dev & 1) == 0 &&
        (nframes == 1 ||
         (bgr_frame_stride % 2 == 0 && mask_frame_stride % 2 == 0));
    if (fast) {
        const long pairs = N / 2;
        const unsigned blocks = (unsigned)((pairs + 255) / 256);
        // short launches stream the model (HBM-bound), long ones are
        // arithmetic-bound: see mog_batch2_k
        if (nframes <= 4)
            hipLaunchKernelGGL(mog_batch2_k<2>, dim3(blocks), dim3(256), 0, s,
                               m->planes, N, (const uint8_t *)bgr_dev,
                               bgr_frame_stride, (uint8_t *)mask_dev,
                               mask_frame_stride, nframes, m->nframes,
                               m->alpha_table, m->c);
        else
            hipLaunchKernelGGL(mog_batch2_k<4>, dim3(blocks), dim3(256), 0, s,
                               m->planes, N, (const uint8_t *)bgr_dev,
                               bgr_frame_stride, (uint8_t *)mask_dev,
                               mask_frame_stride, nframes, m->nframes,
                               m->alpha_table, m->c);
        HK_LAUNCH_CHECK("mog_batch2_k");
    } else {
        const unsigned blocks = (unsigned)((N + 255) / 256);
        hipLaunchKernelGGL(mog_batch1_k, dim3(blocks), dim3(256), 0, s,
                           m->planes, N, m->width, (const uint8_t *)bgr_dev,
                           bgr_frame_stride, bgr_pitch, (uint8_t *)mask_dev,
                           mask_frame_stride, mask_pitch, nframes, m->nframes,
                           m->alpha_table, m->c);
        HK_LAUNCH_CHECK("mog_batch1_k");
    }
    m->nframes += nframes;
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
