// AVX2 forms of the three member / frontier loops of step 3's growth step
// (hk_arcs.cpp).  Same IEEE operations in the same order per value as the
// scalar code -- no FMA, sums accumulated member by member -- so results are
// identical bit for bit; only independent values share an instruction.
// Built without -x hip (host only, -mavx2); callers check simd_available().
#pragma once

#include <cstddef>

namespace hk {

bool simd_available();

// q[i] = squared distance of (f[i], x[i], y[i]) to the arc p = (a, b, c, e,
// f_peak) under the rotation (rot_s, rot_c): arc_dist_sq5() for n tags.
void simd_dist_sq(const double *p, double rot_s, double rot_c, const double *f,
                  const double *x, const double *y, size_t n, double *q);

// True iff some tag is farther from the arc than maxd, i.e. sqrt(q) > maxd,
// decided on q against maxd_sq_lo / maxd_sq_hi and by the root only between.
bool simd_any_farther(const double *p, double rot_s, double rot_c,
                      const double *f, const double *x, const double *y,
                      size_t n, double maxd, double maxd_sq_lo,
                      double maxd_sq_hi);

// The 11 unit-weight moment sums over the tags in order:
// out = T0 T1 T2 T3 T4 X1 TX T2X Y1 TY T2Y  (t = f[i] - f_peak).
void simd_moments(const double *f, const double *x, const double *y, size_t n,
                  double f_peak, double *out11);

}  // namespace hk
