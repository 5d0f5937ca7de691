// H1 -- synthetic juggling video rendered on the device.
//
// The reference ships no clips (its sample videos are external downloads,
// reference README.md:25-27), so benchmark and test frames are synthesised:
// a smooth gradient background with a fixed per-pixel texture, an optional
// slow low-frequency brightness drift, a dark moving "juggler" rectangle,
// anti-aliased discs (balls and distractor blobs, positions supplied per
// frame by the host from hawkeye_amd/synth.py) and per-pixel Gaussian sensor
// noise from a counter-based hash, all in BGR uint8.  Deterministic for a
// given (seed, frame index, pixel), independent of launch geometry.
//
// This is input generation, not part of the timed hot path.

#include <cstdint>

#include "hk_common.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t h) {          // (0, 1]
    return ((float)(h >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

struct RenderCfg {
    int width, height;
    float fps;
    uint32_t seed;
    float noise_sigma;
    float drift_amp, drift_hz;
    int juggler;
};

__global__ __launch_bounds__(256) void render_k(
    const RenderCfg cfg, const float4 *__restrict__ discs, int max_discs,
    int first_frame, int nframes, uint8_t *__restrict__ out, long frame_stride,
    long pitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int t = blockIdx.z;
    if (x >= cfg.width || t >= nframes) return;
    const int frame = first_frame + t;
    const float fx = (float)x, fy = (float)y;
    const float W = (float)cfg.width, H = (float)cfg.height;

    // background: gradient + fixed texture (+ drift)
    const uint32_t ph = mix32((uint32_t)(y * cfg.width + x) * 2654435761u ^
                              cfg.seed);
    const float tex = ((float)(ph & 255u) - 127.5f) * (10.0f / 255.0f);
    float drift = 0.f;
    if (cfg.drift_amp != 0.f)
        drift = cfg.drift_amp *
                __sinf(6.2831853f * cfg.drift_hz * (float)frame / cfg.fps +
                       fx * (4.0f / W) + fy * (3.0f / H));
    float c0 = 95.f + 40.f * fx / W + 25.f * fy / H + tex + drift;     // B
    float c1 = 110.f + 30.f * fy / H + tex + drift;                    // G
    float c2 = 120.f - 35.f * fx / W + 20.f * fy / H + tex + drift;    // R

    if (cfg.juggler) {      // dark body swaying slowly at the bottom centre
        const float jx = 0.5f * W +
                         0.02f * W * __sinf(6.2831853f * 0.3f * (float)frame /
                                            cfg.fps);
        if (fabsf(fx - jx) < 0.07f * W && fy > 0.62f * H) {
            c0 = 38.f + tex; c1 = 35.f + tex; c2 = 40.f + tex;
        }
    }

    // discs: (x, y, radius, kind); kind 0 = ball, 1.. = distractor shades
    const float4 *row = discs + (long)frame * max_discs;
    for (int i = 0; i < max_discs; ++i) {
        const float4 d = row[i];
        if (d.z <= 0.f) continue;
        const float dx = fx - d.x, dy = fy - d.y;
        if (fabsf(dx) > d.z + 1.f || fabsf(dy) > d.z + 1.f) continue;
        const float cov = fminf(fmaxf(d.z + 0.5f - sqrtf(dx * dx + dy * dy),
                                      0.f), 1.f);
        if (cov <= 0.f) continue;
        float b, g, r;
        const int kind = (int)d.w;
        if (kind == 0)      { b = 60.f;  g = 215.f; r = 245.f; }
        else if (kind == 1) { b = 200.f; g = 190.f; r = 60.f; }
        else if (kind == 2) { b = 30.f;  g = 40.f;  r = 180.f; }
        else                { b = 225.f; g = 225.f; r = 225.f; }
        c0 += cov * (b - c0); c1 += cov * (g - c1); c2 += cov * (r - c2);
    }

    // sensor noise (Box-Muller on hashed counters)
    if (cfg.noise_sigma > 0.f) {
        const uint32_t k = mix32(cfg.seed ^ (uint32_t)frame * 0x9E3779B9u) ^
                           (uint32_t)(y * cfg.width + x);
        const uint32_t h0 = mix32(k), h1 = mix32(k ^ 0x68bc21ebu),
                       h2 = mix32(k ^ 0x02e5be93u), h3 = mix32(k ^ 0xa54ff53au);
        const float r0 = sqrtf(-2.f * __logf(u01(h0))),
                    r1 = sqrtf(-2.f * __logf(u01(h2)));
        const float a0 = 6.2831853f * u01(h1), a1 = 6.2831853f * u01(h3);
        c0 += cfg.noise_sigma * r0 * __cosf(a0);
        c1 += cfg.noise_sigma * r0 * __sinf(a0);
        c2 += cfg.noise_sigma * r1 * __cosf(a1);
    }

    uint8_t *px = out + (long)t * frame_stride + (long)y * pitch + 3L * x;
    px[0] = (uint8_t)__float2int_rn(fminf(fmaxf(c0, 0.f), 255.f));
    px[1] = (uint8_t)__float2int_rn(fminf(fmaxf(c1, 0.f), 255.f));
    px[2] = (uint8_t)__float2int_rn(fminf(fmaxf(c2, 0.f), 255.f));
}

}  // namespace

extern "C" int hk_render_frames(int width, int height, float fps,
                                unsigned seed, float noise_sigma,
                                float drift_amp, float drift_hz, int juggler,
                                const float *discs_dev, int max_discs,
                                int first_frame, int nframes, void *bgr_dev,
                                long frame_stride, long pitch, void *stream) {
    if (width <= 0 || height <= 0 || nframes <= 0 || !bgr_dev ||
        (max_discs > 0 && !discs_dev) || pitch < 3L * width || !(fps > 0.f))
        return hk::fail(HK_ERR_ARG, "hk_render_frames: bad argument");
    if (height > 65535 || nframes > 65535)
        return hk::fail(HK_ERR_ARG,
                        "hk_render_frames: at most 65535 rows/frames a call");
    const RenderCfg cfg{width, height, fps, seed, noise_sigma, drift_amp,
                        drift_hz, juggler};
    const dim3 grid((unsigned)((width + 255) / 256), (unsigned)height,
                    (unsigned)nframes);
    hipLaunchKernelGGL(render_k, grid, dim3(256), 0, (hipStream_t)stream, cfg,
                       (const float4 *)discs_dev, max_discs, first_frame,
                       nframes, (uint8_t *)bgr_dev, frame_stride, pitch);
    HK_LAUNCH_CHECK("render_k");
    return HK_OK;
}
