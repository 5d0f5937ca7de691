// Scratch probe (not part of the product): how often do the pixels of the
// batch kernel's waves leave the "component 0 matched" case?
#include "../../hawkeye_amd/csrc/hk_mog.hip"

namespace {
__global__ __launch_bounds__(256) void mog_stats_k(
    float *__restrict__ planes, long n_pixels, const uint8_t *__restrict__ bgr,
    long bgr_frame_stride, int nframes, int frames_before,
    const float *__restrict__ alpha_table, const MogConst c,
    unsigned long long *__restrict__ stats /* [nframes][4] */) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i0 = g * 2;
    if (i0 >= n_pixels) return;
    Mix px[2];
    for (int q = 0; q < 8; ++q)
        for (int k = 0; k < K; ++k) {
            px[0].f[q][k] = planes[(long)(q * K + k) * n_pixels + i0];
            px[1].f[q][k] = planes[(long)(q * K + k) * n_pixels + i0 + 1];
        }
    for (int t = 0; t < nframes; ++t) {
        const uint8_t *s = bgr + (long)t * bgr_frame_stride + i0 * 3;
        int n = frames_before + t + 1;
        n = n < HISTORY ? n : HISTORY;
        const float alpha = alpha_table[n - 1];
        int notok = 0, ncomp = 0;
        for (int j = 0; j < 2; ++j) {
            const float x0 = s[3 * j], x1 = s[3 * j + 1], x2 = s[3 * j + 2];
            const float d0 = x0 - px[j].f[F_M0][0], d1 = x1 - px[j].f[F_M1][0],
                        d2 = x2 - px[j].f[F_M2][0];
            const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
            const bool ok = px[j].f[F_W][0] >= FLT_EPSILON &&
                            dist2 < c.vT * (px[j].f[F_V0][0] + px[j].f[F_V1][0] +
                                            px[j].f[F_V2][0]);
            notok += !ok;
            for (int k = 0; k < K; ++k) ncomp += px[j].f[F_W][k] > 0.f;
            mog_step(px[j], x0, x1, x2, alpha, c);
        }
        const bool lane_mixed = notok != 0;
        const unsigned long long b = __ballot(lane_mixed);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&stats[t * 4 + 0], 1ull);               // waves
            atomicAdd(&stats[t * 4 + 1], b ? 1ull : 0ull);     // mixed waves
        }
        if (notok) atomicAdd(&stats[t * 4 + 2], (unsigned long long)notok);
        atomicAdd(&stats[t * 4 + 3], (unsigned long long)ncomp);
    }
    for (int q = 0; q < 8; ++q)
        for (int k = 0; k < K; ++k) {
            planes[(long)(q * K + k) * n_pixels + i0] = px[0].f[q][k];
            planes[(long)(q * K + k) * n_pixels + i0 + 1] = px[1].f[q][k];
        }
}
}  // namespace

extern "C" int mog_stats(float *planes, long n_pixels, const void *bgr,
                         long stride, int nframes, int frames_before,
                         const float *alpha_table, void *stats) {
    MogConst c;
    const double noiseSigma = 15.;
    c.T = 0.7f; c.vT = 6.25f; c.w0 = 0.05f;
    c.sk0 = (float)(c.w0 / (noiseSigma * 2 * std::sqrt(3.)));
    c.var0 = 900.f; c.minVar = 225.f;
    const unsigned blocks = (unsigned)((n_pixels / 2 + 255) / 256);
    hipLaunchKernelGGL(mog_stats_k, dim3(blocks), dim3(256), 0, 0, planes,
                       n_pixels, (const uint8_t *)bgr, stride, nframes,
                       frames_before, alpha_table, c,
                       (unsigned long long *)stats);
    return (int)hipDeviceSynchronize();
}
