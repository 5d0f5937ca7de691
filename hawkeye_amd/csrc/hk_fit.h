// Parabola-fit arithmetic shared by host code (hk_arcs.cpp) and device code
// (hk_em.hip): the weighted moment sums, the closed-form solve of fit_arcs
// (reference tracker.py:713-798) and the arc-to-tag distance (reference
// types.py:59-72).  One source for both sides, compiled without FMA
// contraction and with IEEE fp64 divide / sqrt on both, so a trial fit gives
// the same bits wherever it runs.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#define HK_HD __host__ __device__
#else
#define HK_HD
#endif

namespace hk {

struct Moments {
    double T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0;
    double X1 = 0, TX = 0, T2X = 0, Y1 = 0, TY = 0, T2Y = 0;
};

// One tag at t = frame - f_peak with weight w.
HK_HD inline void moments_add(Moments &m, double t, double w, double x,
                              double y) {
    const double t2 = t * t;
    m.T0 += w;
    m.T1 += w * t;
    m.T2 += w * t2;
    m.T3 += w * (t2 * t);
    m.T4 += w * (t2 * t2);
    m.X1 += w * x;
    m.TX += w * t * x;
    m.T2X += w * t2 * x;
    m.Y1 += w * y;
    m.TY += w * t * y;
    m.T2Y += w * t2 * y;
}

// p = (a, b, c, e, f_peak).  Leaves p untouched (returns false) if the normal
// equations are near singular or the solution is unusable.
HK_HD inline bool solve_fit5(double *p, const Moments &m, double rot_s,
                             double rot_c) {
    const double T0 = m.T0, T1 = m.T1, T2 = m.T2, T3 = m.T3, T4 = m.T4;
    const double det2 = T0 * T2 - T1 * T1;
    const double det3 = 2 * T1 * T2 * T3 + T0 * T2 * T4 - T0 * (T3 * T3) -
                        (T1 * T1) * T4 - T2 * T2 * T2;
    if (fabs(det2) < 1e-3 || fabs(det3) < 1e-3) return false;
    const double s = rot_s, cs = rot_c;
    const double bx1 = cs * m.TX - s * m.TY;
    const double bx2 = cs * m.X1 - s * m.Y1;
    const double b_new = (T0 * bx1 - T1 * bx2) / det2;
    double a_new = (-T1 * bx1 + T2 * bx2) / det2;
    const double m11 = T0 * T2 - T1 * T1;
    const double m12 = T1 * T2 - T0 * T3;
    const double m13 = T1 * T3 - T2 * T2;
    const double m22 = T0 * T4 - T2 * T2;
    const double m23 = T2 * T3 - T1 * T4;
    const double m33 = T2 * T4 - T3 * T3;
    const double by1 = cs * m.T2Y + s * m.T2X;
    const double by2 = cs * m.TY + s * m.TX;
    const double by3 = cs * m.Y1 + s * m.X1;
    const double e_new = (m11 * by1 + m12 * by2 + m13 * by3) / det3;
    const double d_new = (m12 * by1 + m22 * by2 + m23 * by3) / det3;
    double c_new = (m13 * by1 + m23 * by2 + m33 * by3) / det3;
    if (a_new != a_new || b_new != b_new || c_new != c_new ||
        d_new != d_new || e_new != e_new || e_new == 0)
        return false;
    const double shift = -d_new / (2.0 * e_new);
    p[4] += shift;
    a_new += b_new * shift;
    c_new += d_new * shift + e_new * (shift * shift);
    p[0] = a_new;
    p[1] = b_new;
    p[2] = c_new;
    p[3] = e_new;
    return true;
}

// Squared distance of a tag at (frame, x, y) to the arc p.
HK_HD inline double arc_dist_sq5(const double *p, double frame, double x,
                                 double y, double rot_s, double rot_c) {
    const double t = frame - p[4];
    const double xj = p[0] + p[1] * t;
    const double yj = p[2] + p[3] * t * t;
    const double px = xj * rot_c + yj * rot_s;
    const double py = yj * rot_c - xj * rot_s;
    const double dx = px - x, dy = py - y;
    return dx * dx + dy * dy;
}

// One merge_arcs() trial (reference tracker.py:1199-1233) on a pair of arcs.
struct MergeTrial {
    double p[5];        // the fitted trial arc (arc1's parameters if no fit)
    int union_n;        // tags in the union of the two good-tag lists
    int poor;           // tags farther than max_distance, counted up to 3
};

}  // namespace hk
