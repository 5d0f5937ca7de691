// See hk_simd.h.  Compiled with -mavx2 -ffp-contract=off, never -mfma.
#include "hk_simd.h"

#include <cmath>
#include <immintrin.h>

namespace hk {

bool simd_available() {
    static const bool ok = __builtin_cpu_supports("avx2");
    return ok;
}

namespace {

// arc_dist_sq5 for 4 tags
inline __m256d dist_sq4(__m256d a, __m256d b, __m256d c, __m256d e,
                        __m256d fp, __m256d rs, __m256d rc, __m256d f,
                        __m256d x, __m256d y) {
    const __m256d t = _mm256_sub_pd(f, fp);
    const __m256d xj = _mm256_add_pd(a, _mm256_mul_pd(b, t));
    const __m256d yj =
        _mm256_add_pd(c, _mm256_mul_pd(_mm256_mul_pd(e, t), t));
    const __m256d px =
        _mm256_add_pd(_mm256_mul_pd(xj, rc), _mm256_mul_pd(yj, rs));
    const __m256d py =
        _mm256_sub_pd(_mm256_mul_pd(yj, rc), _mm256_mul_pd(xj, rs));
    const __m256d dx = _mm256_sub_pd(px, x), dy = _mm256_sub_pd(py, y);
    return _mm256_add_pd(_mm256_mul_pd(dx, dx), _mm256_mul_pd(dy, dy));
}

inline double dist_sq1(const double *p, double rs, double rc, double f,
                       double x, double y) {
    const double t = f - p[4];
    const double xj = p[0] + p[1] * t;
    const double yj = p[2] + p[3] * t * t;
    const double px = xj * rc + yj * rs;
    const double py = yj * rc - xj * rs;
    const double dx = px - x, dy = py - y;
    return dx * dx + dy * dy;
}

}  // namespace

void simd_dist_sq(const double *p, double rot_s, double rot_c, const double *f,
                  const double *x, const double *y, size_t n, double *q) {
    const __m256d a = _mm256_set1_pd(p[0]), b = _mm256_set1_pd(p[1]),
                  c = _mm256_set1_pd(p[2]), e = _mm256_set1_pd(p[3]),
                  fp = _mm256_set1_pd(p[4]), rs = _mm256_set1_pd(rot_s),
                  rc = _mm256_set1_pd(rot_c);
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        _mm256_storeu_pd(q + i, dist_sq4(a, b, c, e, fp, rs, rc,
                                         _mm256_loadu_pd(f + i),
                                         _mm256_loadu_pd(x + i),
                                         _mm256_loadu_pd(y + i)));
    for (; i < n; ++i) q[i] = dist_sq1(p, rot_s, rot_c, f[i], x[i], y[i]);
}

bool simd_any_farther(const double *p, double rot_s, double rot_c,
                      const double *f, const double *x, const double *y,
                      size_t n, double maxd, double lo, double hi) {
    const __m256d a = _mm256_set1_pd(p[0]), b = _mm256_set1_pd(p[1]),
                  c = _mm256_set1_pd(p[2]), e = _mm256_set1_pd(p[3]),
                  fp = _mm256_set1_pd(p[4]), rs = _mm256_set1_pd(rot_s),
                  rc = _mm256_set1_pd(rot_c), vlo = _mm256_set1_pd(lo),
                  vhi = _mm256_set1_pd(hi);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const __m256d q = dist_sq4(a, b, c, e, fp, rs, rc,
                                   _mm256_loadu_pd(f + i),
                                   _mm256_loadu_pd(x + i),
                                   _mm256_loadu_pd(y + i));
        if (_mm256_movemask_pd(_mm256_cmp_pd(q, vhi, _CMP_GT_OQ))) return true;
        // not clearly inside (q < lo false, NaN included): take the root
        const int unsure = _mm256_movemask_pd(_mm256_cmp_pd(q, vlo, _CMP_NLT_UQ));
        if (unsure) {
            alignas(32) double qs[4];
            _mm256_store_pd(qs, q);
            for (int k = 0; k < 4; ++k)
                if (((unsure >> k) & 1) && std::sqrt(qs[k]) > maxd) return true;
        }
    }
    for (; i < n; ++i) {
        const double q = dist_sq1(p, rot_s, rot_c, f[i], x[i], y[i]);
        if (q > hi) return true;
        if (q < lo) continue;
        if (std::sqrt(q) > maxd) return true;
    }
    return false;
}

namespace {
// rows r0..r3 (4 members x one value each) -> m0..m3 (one member x 4 values)
inline void transpose4(__m256d r0, __m256d r1, __m256d r2, __m256d r3,
                       __m256d &m0, __m256d &m1, __m256d &m2, __m256d &m3) {
    const __m256d t0 = _mm256_unpacklo_pd(r0, r1);   // r0[0] r1[0] r0[2] r1[2]
    const __m256d t1 = _mm256_unpackhi_pd(r0, r1);   // r0[1] r1[1] r0[3] r1[3]
    const __m256d t2 = _mm256_unpacklo_pd(r2, r3);
    const __m256d t3 = _mm256_unpackhi_pd(r2, r3);
    m0 = _mm256_permute2f128_pd(t0, t2, 0x20);
    m1 = _mm256_permute2f128_pd(t1, t3, 0x20);
    m2 = _mm256_permute2f128_pd(t0, t2, 0x31);
    m3 = _mm256_permute2f128_pd(t1, t3, 0x31);
}
}  // namespace

void simd_moments(const double *f, const double *x, const double *y, size_t n,
                  double f_peak, double *out) {
    // acc0 = T0 T1 T2 T3, acc1 = T4 X1 TX T2X, acc2 = Y1 TY T2Y -
    __m256d acc0 = _mm256_setzero_pd(), acc1 = acc0, acc2 = acc0;
    const __m256d fp = _mm256_set1_pd(f_peak);
    const __m256d one = _mm256_set1_pd(1.0), zero = _mm256_setzero_pd();
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        const __m256d t = _mm256_sub_pd(_mm256_loadu_pd(f + i), fp);
        const __m256d vx = _mm256_loadu_pd(x + i), vy = _mm256_loadu_pd(y + i);
        const __m256d t2 = _mm256_mul_pd(t, t);
        const __m256d t3 = _mm256_mul_pd(t2, t);
        const __m256d t4 = _mm256_mul_pd(t2, t2);
        const __m256d tx = _mm256_mul_pd(t, vx);
        const __m256d t2x = _mm256_mul_pd(t2, vx);
        const __m256d ty = _mm256_mul_pd(t, vy);
        const __m256d t2y = _mm256_mul_pd(t2, vy);
        __m256d a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
        transpose4(one, t, t2, t3, a0, a1, a2, a3);
        transpose4(t4, vx, tx, t2x, b0, b1, b2, b3);
        transpose4(vy, ty, t2y, zero, c0, c1, c2, c3);
        acc0 = _mm256_add_pd(acc0, a0);
        acc1 = _mm256_add_pd(acc1, b0);
        acc2 = _mm256_add_pd(acc2, c0);
        acc0 = _mm256_add_pd(acc0, a1);
        acc1 = _mm256_add_pd(acc1, b1);
        acc2 = _mm256_add_pd(acc2, c1);
        acc0 = _mm256_add_pd(acc0, a2);
        acc1 = _mm256_add_pd(acc1, b2);
        acc2 = _mm256_add_pd(acc2, c2);
        acc0 = _mm256_add_pd(acc0, a3);
        acc1 = _mm256_add_pd(acc1, b3);
        acc2 = _mm256_add_pd(acc2, c3);
    }
    alignas(32) double s[12];
    _mm256_store_pd(s, acc0);
    _mm256_store_pd(s + 4, acc1);
    _mm256_store_pd(s + 8, acc2);
    for (; i < n; ++i) {
        const double t = f[i] - f_peak;
        const double t2 = t * t;
        s[0] += 1.0;
        s[1] += t;
        s[2] += t2;
        s[3] += t2 * t;
        s[4] += t2 * t2;
        s[5] += x[i];
        s[6] += t * x[i];
        s[7] += t2 * x[i];
        s[8] += y[i];
        s[9] += t * y[i];
        s[10] += t2 * y[i];
    }
    for (int k = 0; k < 11; ++k) out[k] = s[k];
}

}  // namespace hk
