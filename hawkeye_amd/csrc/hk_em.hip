// K5 / K6 / K7 -- EM arc fitting kernels for gfx950, all fp64.
//
//   K5  E-step  (reference tracker.py:992-1039, calculate_weights)
//       estep_raw_k    one workgroup per arc sweeps the arc's frame window
//                      (a contiguous range of tag indices, tags being sorted
//                      by frame): radius gate, squared distance / sigma^2,
//                      exp(-q) for q < 5 -> dense window of raw weights
//       estep_total_k  one lane per tag: sum of its raw weights over arcs in
//                      arc-list order (the reference's dict order)
//       estep_norm_k   per arc: w * total_weight / sum, membership bits by
//                      wave ballot
//   K6  M-step  (reference tracker.py:698-798 fit_arcs, 1041-1124 tilt)
//       arc_moments_k  one wavefront per arc: 11 weighted moment sums reduced
//                      across the wave by shuffles, lane 0 solves the 2x2 and
//                      3x3 normal equations by adjugate and recentres f_peak
//   K7  distances (reference types.py:59-72 over an arc's tags)
//       member_dist_k  one wavefront per arc over its explicit member list
//
// The data are tiny (10^4-10^5 tags, 10^2 arcs): these launches are latency
// bound, not bandwidth bound; no roofline fraction is claimed for them.
// No FMA contraction, IEEE fp64 divide / sqrt, so the per-element arithmetic
// equals the host's; only summation order differs from the oracle.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "hk_clip.h"

namespace {

struct TagArrays {
    const int *frame;
    const double *x, *y, *r, *tw;
};

// arc_params row: a b c e f_peak r_lo r_hi -
// arc_win row:    lo hi w_off bitmap_off     (dense: tag window [lo, hi))
//                 lo hi -1    -              (list: entries [lo, hi) of list)
constexpr int AP = 8, AW = 4;

__device__ __forceinline__ void position(const double *__restrict__ p,
                                         double frame, double sn, double cs,
                                         double &px, double &py) {
    const double t = frame - p[4];
    const double xj = p[0] + p[1] * t;
    const double yj = p[2] + p[3] * t * t;
    px = xj * cs + yj * sn;
    py = yj * cs - xj * sn;
}

__global__ __launch_bounds__(256) void estep_raw_k(
    const TagArrays tg, const double *__restrict__ arc_params,
    const long *__restrict__ arc_win, double *__restrict__ weights,
    const int *__restrict__ arc_clip, const hk::ClipDev *__restrict__ clip) {
    const int a = blockIdx.x;
    const double sn = clip[arc_clip[a]].sn, cs = clip[arc_clip[a]].cs;
    const double sigmasq = clip[arc_clip[a]].sigmasq;
    const double *p = arc_params + (long)a * AP;
    const long lo = arc_win[a * AW], hi = arc_win[a * AW + 1];
    const long off = arc_win[a * AW + 2];
    const double r_lo = p[5], r_hi = p[6];
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        double w = 0.0;
        const double r = tg.r[i];
        if (r_lo <= r && r <= r_hi) {
            double px, py;
            position(p, (double)tg.frame[i], sn, cs, px, py);
            const double dx = px - tg.x[i], dy = py - tg.y[i];
            const double q = (dx * dx + dy * dy) / sigmasq;
            if (q < 5.0) w = exp(-q);
        }
        weights[off + (i - lo)] = w;
    }
}

// grid: (blocks of 256 tags, clips of the launch)
__global__ __launch_bounds__(256) void estep_total_k(
    const hk::ClipDev *__restrict__ clip, const long *__restrict__ arc_win_all,
    const double *__restrict__ weights, double *__restrict__ total) {
    __shared__ long win[256 * 3];
    const hk::ClipDev cd = clip[blockIdx.y];
    if ((int)(blockIdx.x * 256) >= cd.n_tags) return;   // the whole workgroup
    const int n = cd.tag_base + cd.n_tags, narcs = cd.n_arcs;
    const long *arc_win = arc_win_all + (long)cd.arc_base * AW;
    const int i = cd.tag_base + blockIdx.x * 256 + threadIdx.x;
    double sum = 0.0;
    for (int base = 0; base < narcs; base += 256) {
        const int m = min(256, narcs - base);
        __syncthreads();
        if ((int)threadIdx.x < m) {
            const long *w = arc_win + (long)(base + threadIdx.x) * AW;
            win[threadIdx.x * 3] = w[0];
            win[threadIdx.x * 3 + 1] = w[1];
            win[threadIdx.x * 3 + 2] = w[2];
        }
        __syncthreads();
        if (i < n) {
            for (int k = 0; k < m; ++k) {
                const long lo = win[k * 3], hi = win[k * 3 + 1];
                if (i >= lo && i < hi) sum += weights[win[k * 3 + 2] + (i - lo)];
            }
        }
    }
    if (i < n) total[i] = sum;
}

__global__ __launch_bounds__(256) void estep_norm_k(
    const TagArrays tg, const long *__restrict__ arc_win,
    double *__restrict__ weights, const double *__restrict__ total,
    unsigned long long *__restrict__ bitmap) {
    const int a = blockIdx.x;
    const long lo = arc_win[a * AW], hi = arc_win[a * AW + 1];
    const long off = arc_win[a * AW + 2], boff = arc_win[a * AW + 3];
    const long span = hi - lo;
    const long padded = (span + 63) & ~63L;
    for (long s = threadIdx.x; s < padded; s += 256) {
        bool member = false;
        if (s < span) {
            double w = weights[off + s];
            if (w > 0.0) {
                const long i = lo + s;
                const double tot = total[i];
                if (tot > 1e-5) {
                    w = w * tg.tw[i] / tot;
                    member = true;
                } else {
                    w = 0.0;
                }
                weights[off + s] = w;
            }
        }
        const unsigned long long bits = __ballot(member);
        if ((threadIdx.x & 63) == 0) bitmap[boff + (s >> 6)] = bits;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// what == 0: fit (arc_out row: updated a b c e f_peak)
// what == 1: tilt (arc_out row: valid e_x e_y)
__global__ __launch_bounds__(64) void arc_moments_k(
    const TagArrays tg, const double *__restrict__ arc_params,
    const long *__restrict__ arc_win, const double *__restrict__ weights,
    const int *__restrict__ list, double *__restrict__ arc_out,
    const int *__restrict__ arc_clip, const hk::ClipDev *__restrict__ clip,
    int what) {
    const int a = blockIdx.x;
    const int lane = threadIdx.x;
    const double sn = clip[arc_clip[a]].sn, cs = clip[arc_clip[a]].cs;
    const double *p = arc_params + (long)a * AP;
    const long lo = arc_win[a * AW], hi = arc_win[a * AW + 1];
    const long off = arc_win[a * AW + 2];
    const bool dense = off >= 0;
    const double f_peak = p[4];

    double T0 = 0, T1 = 0, T2 = 0, T3 = 0, T4 = 0;
    double X1 = 0, TX = 0, T2X = 0, Y1 = 0, TY = 0, T2Y = 0;
    int count = 0;
    for (long s = lo + lane; s < hi; s += 64) {
        long i;
        double w;
        if (dense) {
            i = s;
            w = weights[off + (s - lo)];
            if (!(w > 0.0)) continue;
        } else {
            i = list[s];
            w = 1.0;
        }
        ++count;
        const double t = (double)tg.frame[i] - f_peak;
        const double x = tg.x[i], y = tg.y[i];
        const double t2 = t * t;
        T0 += w;
        T1 += w * t;
        T2 += w * t2;
        T3 += w * (t2 * t);
        T4 += w * (t2 * t2);
        X1 += w * x;
        TX += w * t * x;
        T2X += w * t2 * x;
        Y1 += w * y;
        TY += w * t * y;
        T2Y += w * t2 * y;
    }
    T0 = wave_sum(T0); T1 = wave_sum(T1); T2 = wave_sum(T2);
    T3 = wave_sum(T3); T4 = wave_sum(T4);
    X1 = wave_sum(X1); TX = wave_sum(TX); T2X = wave_sum(T2X);
    Y1 = wave_sum(Y1); TY = wave_sum(TY); T2Y = wave_sum(T2Y);
    int total_count = count;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) total_count += __shfl_xor(total_count, d, 64);
    if (lane != 0) return;

    double *out = arc_out + (long)a * AP;
    out[0] = 0.0;                       // "nothing to report"
    if (total_count < 3) return;
    const double det3 = 2 * T1 * T2 * T3 + T0 * T2 * T4 - T0 * (T3 * T3) -
                        (T1 * T1) * T4 - T2 * T2 * T2;
    const double m11 = T0 * T2 - T1 * T1;
    const double m12 = T1 * T2 - T0 * T3;
    const double m13 = T1 * T3 - T2 * T2;
    if (what == 1) {
        if (fabs(det3) < 1e-3) return;
        out[0] = 1.0;
        out[1] = (m11 * T2X + m12 * TX + m13 * X1) / det3;      // e_x
        out[2] = (m11 * T2Y + m12 * TY + m13 * Y1) / det3;      // e_y
        return;
    }
    const double det2 = T0 * T2 - T1 * T1;
    if (fabs(det2) < 1e-3 || fabs(det3) < 1e-3) return;
    const double bx1 = cs * TX - sn * TY;
    const double bx2 = cs * X1 - sn * Y1;
    const double b_new = (T0 * bx1 - T1 * bx2) / det2;
    double a_new = (-T1 * bx1 + T2 * bx2) / det2;
    const double m22 = T0 * T4 - T2 * T2;
    const double m23 = T2 * T3 - T1 * T4;
    const double m33 = T2 * T4 - T3 * T3;
    const double by1 = cs * T2Y + sn * T2X;
    const double by2 = cs * TY + sn * TX;
    const double by3 = cs * Y1 + sn * X1;
    const double e_new = (m11 * by1 + m12 * by2 + m13 * by3) / det3;
    const double d_new = (m12 * by1 + m22 * by2 + m23 * by3) / det3;
    double c_new = (m13 * by1 + m23 * by2 + m33 * by3) / det3;
    if (isnan(a_new) || isnan(b_new) || isnan(c_new) || isnan(d_new) ||
        isnan(e_new) || e_new == 0.0)
        return;
    const double shift = -d_new / (2.0 * e_new);
    a_new += b_new * shift;
    c_new += d_new * shift + e_new * (shift * shift);
    out[0] = 1.0;
    out[1] = a_new;
    out[2] = b_new;
    out[3] = c_new;
    out[4] = e_new;
    out[5] = f_peak + shift;
}

__global__ __launch_bounds__(64) void member_dist_k(
    const TagArrays tg, const double *__restrict__ arc_params,
    const long *__restrict__ arc_win, const int *__restrict__ list,
    double *__restrict__ list_out, const int *__restrict__ arc_clip,
    const hk::ClipDev *__restrict__ clip) {
    const int a = blockIdx.x;
    const double sn = clip[arc_clip[a]].sn, cs = clip[arc_clip[a]].cs;
    const double *p = arc_params + (long)a * AP;
    const long lo = arc_win[a * AW], hi = arc_win[a * AW + 1];
    for (long s = lo + threadIdx.x; s < hi; s += 64) {
        const int i = list[s];
        double px, py;
        position(p, (double)tg.frame[i], sn, cs, px, py);
        const double dx = px - tg.x[i], dy = py - tg.y[i];
        list_out[s] = sqrt(dx * dx + dy * dy);
    }
}

// K8 merge trials: one lane per candidate pair of arcs (reference
// tracker.py:1199-1233).  The lane walks the union of the two arcs' good-tag
// lists in ascending tag order -- the order, and through hk_fit.h the
// operations, of the host's own trial fit, so the result is the same bit for
// bit -- fits the trial parabola and counts the tags it leaves farther than
// max_distance_pixels (up to 3).  Latency-bound: ~10^3-10^4 pairs of ~10^2
// tags; what it buys is all trial fits of a merge pass in one launch.
// (arcs, good-tag lists and pairs of all clips of the launch end to end:
// pairs hold rows of params5 / good_off, the lists tags of the batch)
__global__ __launch_bounds__(64) void merge_trials_k(
    TagArrays tg, const double *__restrict__ params5,
    const int *__restrict__ good_off, const int *__restrict__ good,
    const int *__restrict__ pairs, int npairs, hk::MergeTrial *__restrict__ out,
    const int *__restrict__ arc_clip, const hk::ClipDev *__restrict__ clip) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= npairs) return;
    const int i1 = pairs[2 * q], i2 = pairs[2 * q + 1];
    const hk::ClipDev cd = clip[arc_clip[i1]];
    const double sn = cd.sn, cs = cd.cs, maxd = cd.maxd;
    const int min_tags = cd.min_tags;
    const int lo1 = good_off[i1], hi1 = good_off[i1 + 1];
    const int lo2 = good_off[i2], hi2 = good_off[i2 + 1];
    hk::MergeTrial r;
#pragma unroll
    for (int j = 0; j < 5; ++j) r.p[j] = params5[5 * i1 + j];
    r.poor = 0;

    hk::Moments m;
    int n = 0;
    {
        const double fp = r.p[4];
        int a = lo1, b = lo2;
        while (a < hi1 || b < hi2) {
            const int ta = a < hi1 ? good[a] : 0x7fffffff;
            const int tb = b < hi2 ? good[b] : 0x7fffffff;
            const int t = ta < tb ? ta : tb;
            a += (ta == t);
            b += (tb == t);
            hk::moments_add(m, tg.frame[t] - fp, tg.tw[t], tg.x[t], tg.y[t]);
            ++n;
        }
    }
    r.union_n = n;
    if (n >= min_tags) {
        if (n >= 3) hk::solve_fit5(r.p, m, sn, cs);
        int a = lo1, b = lo2, poor = 0;
        while ((a < hi1 || b < hi2) && poor <= 2) {
            const int ta = a < hi1 ? good[a] : 0x7fffffff;
            const int tb = b < hi2 ? good[b] : 0x7fffffff;
            const int t = ta < tb ? ta : tb;
            a += (ta == t);
            b += (tb == t);
            const double d = sqrt(hk::arc_dist_sq5(r.p, (double)tg.frame[t],
                                                   tg.x[t], tg.y[t], sn, cs));
            if (d > maxd) ++poor;
        }
        r.poor = poor;
    }
    out[q] = r;
}

template <class T>
hipError_t dev_alloc(T *&p, size_t count) {
    return hipMalloc((void **)&p, sizeof(T) * (count ? count : 1));
}
template <class T>
hipError_t pin_alloc(T *&p, size_t count) {
    return hipHostMalloc((void **)&p, sizeof(T) * (count ? count : 1));
}

// Waits for the batch's stream.  The thread SLEEPS (an event created for
// blocking waits) instead of spinning as hipStreamSynchronize does: the waits
// of step 4 are 0.1-1 ms each beside the MOG launches of a batch, a dozen per
// clip, and the host cores are what bounds a batch -- a spinning wait takes
// a core from a step-3 build.  HK_EM_SPIN=1: hipStreamSynchronize.
int wait_for_stream(hk::EmBatch &b) {
    static const bool spin = getenv("HK_EM_SPIN") != nullptr;
    hk::DeviceState &d = b.dev;
    if (spin) {
        HK_HIP(hipStreamSynchronize(b.stream));
        return HK_OK;
    }
    if (!d.waited) {
        hipEvent_t ev = nullptr;
        HK_HIP(hipEventCreateWithFlags(
            &ev, hipEventDisableTiming | hipEventBlockingSync));
        d.waited = ev;
    }
    HK_HIP(hipEventRecord((hipEvent_t)d.waited, b.stream));
    HK_HIP(hipEventSynchronize((hipEvent_t)d.waited));
    return HK_OK;
}

TagArrays tag_arrays(const hk::DeviceState &d) {
    return TagArrays{d.frame, d.x, d.y, d.r, d.tw};
}

// The per-launch tables (arc parameters, windows, member lists, the clip
// table) and the small results (fit / tilt rows, distances, membership bits)
// are NOT copied: they live in page-locked host memory, which the kernels
// read and write directly (h_* and its device-side name are one pointer).  A
// copy each way is a command of its own on the stream -- beside the MOG
// launches of a batch, a copy kernel that has to find wave slots like any
// other: five of them per phase were most of what step 4 cost under load.
// The tables are a few KB; every entry is read once.  (The tags, the dense
// E-step weights and the merge trials' inputs, which are read many times
// over, stay in device memory.)
int ensure_arc_capacity(hk::DeviceState &d, int narcs) {
    if (narcs <= d.arc_cap) return HK_OK;
    const int cap = std::max(256, narcs * 2);
    (void)hipHostFree(d.h_arc_params); (void)hipHostFree(d.h_arc_win);
    (void)hipHostFree(d.h_arc_out); (void)hipHostFree(d.h_arc_clip);
    d.arc_params = nullptr; d.arc_win = nullptr; d.arc_out = nullptr;
    d.arc_clip = nullptr;
    d.h_arc_params = nullptr; d.h_arc_win = nullptr; d.h_arc_out = nullptr;
    d.h_arc_clip = nullptr;
    d.arc_cap = 0;
    HK_HIP(pin_alloc(d.h_arc_params, (size_t)cap * AP));
    HK_HIP(pin_alloc(d.h_arc_win, (size_t)cap * AW));
    HK_HIP(pin_alloc(d.h_arc_out, (size_t)cap * AP));
    HK_HIP(pin_alloc(d.h_arc_clip, (size_t)cap));
    d.arc_params = d.h_arc_params;
    d.arc_win = d.h_arc_win;
    d.arc_out = d.h_arc_out;
    d.arc_clip = d.h_arc_clip;
    d.arc_cap = cap;
    return HK_OK;
}

int ensure_clip_capacity(hk::DeviceState &d, int nclips) {
    if (nclips <= d.clip_cap) return HK_OK;
    const int cap = std::max(16, nclips * 2);
    (void)hipHostFree(d.h_clip_tab);
    d.clip_tab = nullptr;
    d.h_clip_tab = nullptr;
    d.clip_cap = 0;
    HK_HIP(pin_alloc(d.h_clip_tab, (size_t)cap));
    d.clip_tab = d.h_clip_tab;
    d.clip_cap = cap;
    return HK_OK;
}

int ensure_weight_capacity(hk::DeviceState &d, long doubles, long words) {
    if (doubles > d.weight_cap) {
        const long cap = std::max(doubles * 2, 1L << 16);
        (void)hipFree(d.weights);
        d.weights = nullptr;
        d.weight_cap = 0;
        HK_HIP(dev_alloc(d.weights, (size_t)cap));
        d.weight_cap = cap;
    }
    if (words > d.bitmap_cap) {
        const long cap = std::max(words * 2, 1L << 12);
        (void)hipHostFree(d.h_bitmap);
        d.bitmap = nullptr;
        d.h_bitmap = nullptr;
        d.bitmap_cap = 0;
        HK_HIP(pin_alloc(d.h_bitmap, (size_t)cap));
        d.bitmap = d.h_bitmap;
        d.bitmap_cap = cap;
    }
    return HK_OK;
}

int ensure_list_capacity(hk::DeviceState &d, long entries) {
    if (entries <= d.list_cap) return HK_OK;
    const long cap = std::max(entries * 2, 1L << 14);
    (void)hipHostFree(d.h_list); (void)hipHostFree(d.h_list_out);
    d.list = nullptr; d.list_out = nullptr;
    d.h_list = nullptr; d.h_list_out = nullptr;
    d.list_cap = 0;
    HK_HIP(pin_alloc(d.h_list, (size_t)cap));
    HK_HIP(pin_alloc(d.h_list_out, (size_t)cap));
    d.list = d.h_list;
    d.list_out = d.h_list_out;
    d.list_cap = cap;
    return HK_OK;
}

// The clip table and the row ranges of a launch over the clips `act`: clip
// act[j] takes rows [b.arc_base[j], b.arc_base[j] + arcs of that clip).
int stage_clips(hk::EmBatch &b, const std::vector<int> &act) {
    hk::DeviceState &d = b.dev;
    int rc = ensure_clip_capacity(d, (int)act.size());
    if (rc != HK_OK) return rc;
    b.arc_base.assign(act.size(), 0);
    int rows = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        const hk_clip &c = *b.clips[(size_t)act[j]];
        const hk::Rot rot = hk::rotation(c);
        hk::ClipDev &cd = d.h_clip_tab[j];
        cd.sn = rot.s;
        cd.cs = rot.c;
        cd.sigmasq = c.p.sigmasq;
        cd.maxd = c.p.max_distance_pixels;
        cd.tag_base = b.tag_base[(size_t)act[j]];
        cd.n_tags = c.n;
        cd.arc_base = rows;
        cd.n_arcs = (int)c.arcs.size();
        cd.min_tags = c.p.min_tags_per_arc;
        cd.pad = 0;
        b.arc_base[j] = rows;
        rows += cd.n_arcs;
    }
    b.rows = rows;
    rc = ensure_arc_capacity(d, rows);
    if (rc != HK_OK) return rc;
    for (size_t j = 0; j < act.size(); ++j) {
        const int n = d.h_clip_tab[j].n_arcs;
        std::fill(d.h_arc_clip + b.arc_base[j], d.h_arc_clip + b.arc_base[j] + n,
                  (int)j);
    }
    return HK_OK;
}

// Fills the pinned arc tables.  Arcs with dense weights keep their window;
// the others get their explicit member list appended to h_list.  Tags are
// numbered within the batch.
int stage_arcs(hk::EmBatch &b, const std::vector<int> &act,
               long &list_entries) {
    hk::DeviceState &d = b.dev;
    int rc = stage_clips(b, act);
    if (rc != HK_OK) return rc;
    long total = 0;
    for (int ci : act)
        for (const hk::Arc &arc : b.clips[(size_t)ci]->arcs)
            if (arc.w_off < 0) total += (long)arc.tags.size();
    rc = ensure_list_capacity(d, total);
    if (rc != HK_OK) return rc;
    long pos = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        const hk_clip &c = *b.clips[(size_t)act[j]];
        const int base = b.tag_base[(size_t)act[j]];
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            const hk::Arc &arc = c.arcs[k];
            const long row = b.arc_base[j] + (long)k;
            double *p = d.h_arc_params + row * AP;
            p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
            p[4] = arc.f_peak; p[5] = p[6] = p[7] = 0.0;
            long *w = d.h_arc_win + row * AW;
            if (arc.w_off >= 0) {
                w[0] = base + arc.w_lo; w[1] = base + arc.w_hi;
                w[2] = arc.w_off; w[3] = 0;
            } else {
                w[0] = pos;
                for (int i : arc.tags) d.h_list[pos++] = base + i;
                w[1] = pos; w[2] = -1; w[3] = 0;
            }
        }
    }
    list_entries = pos;
    return HK_OK;
}

int launch_moments(hk::EmBatch &b, int what) {
    hk::DeviceState &d = b.dev;
    hipLaunchKernelGGL(arc_moments_k, dim3((unsigned)b.rows), dim3(64), 0,
                       b.stream, tag_arrays(d), d.arc_params, d.arc_win,
                       d.weights, d.list, d.arc_out, d.arc_clip, d.clip_tab,
                       what);
    HK_LAUNCH_CHECK("arc_moments_k");
    return HK_OK;
}

int run_moments(hk::EmBatch &b, const std::vector<int> &act, int what) {
    long entries = 0;
    int rc = stage_arcs(b, act, entries);
    if (rc != HK_OK) return rc;
    if (b.rows == 0) return HK_OK;
    rc = launch_moments(b, what);
    if (rc != HK_OK) return rc;
    { const int src = wait_for_stream(b); if (src != HK_OK) return src; }
    return HK_OK;
}

// arc_out rows of a fit -> the arcs of the clips of the launch
void apply_fit(hk::EmBatch &b, const std::vector<int> &act) {
    for (size_t j = 0; j < act.size(); ++j) {
        hk_clip &c = *b.clips[(size_t)act[j]];
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            const double *o = b.dev.h_arc_out + (b.arc_base[j] + (long)k) * AP;
            if (o[0] == 0.0) continue;
            hk::Arc &arc = c.arcs[k];
            arc.a = o[1]; arc.b = o[2]; arc.c = o[3]; arc.e = o[4];
            arc.f_peak = o[5];
        }
    }
}

}  // namespace

namespace hk {

// Device workspaces outlive the batches that use them: hipMalloc / hipFree
// (the latter synchronises the whole device) cost more than a whole EM pass
// on a clip, so a released workspace is parked here, per device, and the next
// batch on that device takes it over.  hk_release_cached_memory() frees them.
// A workspace of a 50 000-tag clip is a few MB.
namespace {
constexpr size_t POOL_MAX = 48;
std::mutex g_pool_mutex;
std::vector<DeviceState> g_pool;

// The tag uploads of em_open() are asynchronous copies out of the workspace's
// pinned block.  A batch without arcs never synchronises its stream (every EM
// call returns early), and an error path may skip the wait too: before the
// workspace is parked, freed or its block refilled, the copies must have left
// it -- otherwise the next owner, on another stream, could overwrite h_tags
// under them or see its own uploads overtaken by them.
void wait_for_uploads(DeviceState &d) {
    if (!d.upload_pending || !d.uploaded) return;
    DeviceGuard guard(d.device);
    (void)hipEventSynchronize((hipEvent_t)d.uploaded);
    d.upload_pending = false;
}

void free_workspace(DeviceState &d) {
    DeviceGuard guard(d.device);
    wait_for_uploads(d);
    if (d.uploaded) (void)hipEventDestroy((hipEvent_t)d.uploaded);
    if (d.waited) (void)hipEventDestroy((hipEvent_t)d.waited);
    (void)hipFree(d.frame); (void)hipFree(d.x); (void)hipFree(d.y);
    (void)hipFree(d.r); (void)hipFree(d.tw); (void)hipFree(d.total);
    (void)hipHostFree(d.h_tags);
    (void)hipFree(d.weights);
    (void)hipHostFree(d.h_arc_params); (void)hipHostFree(d.h_arc_win);
    (void)hipHostFree(d.h_arc_out); (void)hipHostFree(d.h_arc_clip);
    (void)hipHostFree(d.h_clip_tab); (void)hipHostFree(d.h_bitmap);
    (void)hipHostFree(d.h_list); (void)hipHostFree(d.h_list_out);
    (void)hipFree(d.merge_buf); (void)hipHostFree(d.h_merge_buf);
    d = DeviceState();
}

int ensure_tag_capacity(DeviceState &d, int n) {
    if (n <= d.tag_cap) return HK_OK;
    const size_t cap = (size_t)std::max(2 * (long)n, 1L << 16);
    (void)hipFree(d.frame); (void)hipFree(d.x); (void)hipFree(d.y);
    (void)hipFree(d.r); (void)hipFree(d.tw); (void)hipFree(d.total);
    (void)hipHostFree(d.h_tags);
    d.frame = nullptr;
    d.x = d.y = d.r = d.tw = d.total = nullptr;
    d.h_tags = nullptr;
    d.tag_cap = 0;
    HK_HIP(dev_alloc(d.frame, cap));
    HK_HIP(dev_alloc(d.x, cap));
    HK_HIP(dev_alloc(d.y, cap));
    HK_HIP(dev_alloc(d.r, cap));
    HK_HIP(dev_alloc(d.tw, cap));
    HK_HIP(dev_alloc(d.total, cap));
    HK_HIP(pin_alloc(d.h_tags, cap * (sizeof(int) + 4 * sizeof(double))));
    d.tag_cap = (int)cap;
    return HK_OK;
}

// The tags of all clips of the batch, end to end, on their way to the device.
int upload_tags(EmBatch &b) {
    DeviceState &d = b.dev;
    const hipStream_t s = b.stream;
    const int rc = ensure_tag_capacity(d, d.n);
    if (rc != HK_OK) return rc;
    wait_for_uploads(d);        // (a workspace that em_close parked has none)
    const size_t n = (size_t)d.n;
    if (n == 0) return HK_OK;
    // Through the workspace's pinned block: a copy from the clips' pageable
    // vectors is a synchronous call of ~0.2 ms each that holds up other
    // threads' launches (seen in a HIP API trace of a batch).  The block is
    // the workspace's own; an event behind the copies is waited for before
    // the workspace is handed on (wait_for_uploads).
    if (!d.uploaded) {
        hipEvent_t ev = nullptr;
        HK_HIP(hipEventCreateWithFlags(
            &ev, hipEventDisableTiming | hipEventBlockingSync));
        d.uploaded = ev;
    }
    const size_t cap = (size_t)d.tag_cap;
    double *hx = reinterpret_cast<double *>(d.h_tags);
    double *hy = hx + cap, *hr = hy + cap, *htw = hr + cap;
    int *hf = reinterpret_cast<int *>(htw + cap);
    for (size_t k = 0; k < b.clips.size(); ++k) {
        const hk_clip &c = *b.clips[k];
        const size_t at = (size_t)b.tag_base[k], m = (size_t)c.n;
        if (m == 0) continue;
        std::memcpy(hx + at, c.x.data(), sizeof(double) * m);
        std::memcpy(hy + at, c.y.data(), sizeof(double) * m);
        std::memcpy(hr + at, c.r.data(), sizeof(double) * m);
        std::memcpy(htw + at, c.tw.data(), sizeof(double) * m);
        std::memcpy(hf + at, c.frame.data(), sizeof(int) * m);
    }
    // (pending from the first copy on, so that an error half-way still waits:
    // the event then completes at once or behind what was queued)
    d.upload_pending = true;
    hipError_t e = hipMemcpyAsync(d.frame, hf, sizeof(int) * n,
                                  hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d.x, hx, sizeof(double) * n, hipMemcpyHostToDevice,
                           s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d.y, hy, sizeof(double) * n, hipMemcpyHostToDevice,
                           s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d.r, hr, sizeof(double) * n, hipMemcpyHostToDevice,
                           s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d.tw, htw, sizeof(double) * n,
                           hipMemcpyHostToDevice, s);
    const hipError_t e2 = hipEventRecord((hipEvent_t)d.uploaded, s);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(s);
        d.upload_pending = false;
        HK_HIP(e);
    }
    return HK_OK;
}
}  // namespace

int em_open(hk_clip *const *clips, int n, hipStream_t s, EmBatch **out) {
    *out = nullptr;
    if (n <= 0) return fail(HK_ERR_ARG, "EM batch without clips");
    const int device = clips[0]->device;
    long total = 0;
    for (int k = 0; k < n; ++k) {
        if (clips[k]->device != device)
            return fail(HK_ERR_ARG, "the clips of an EM batch must be on one "
                                    "device (%d and %d)", device,
                        clips[k]->device);
        total += clips[k]->n;
    }
    if (total > (1L << 30))
        return fail(HK_ERR_CAPACITY, "EM batch of %ld tags", total);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(HK_ERR_HIP, "cannot select device %d", device);
    EmBatch *b = new EmBatch();
    b->clips.assign(clips, clips + n);
    b->device = device;
    b->stream = s;
    b->tag_base.resize((size_t)n);
    int at = 0;
    for (int k = 0; k < n; ++k) {
        b->tag_base[(size_t)k] = at;
        at += clips[k]->n;
    }
    {                                       // take over a parked workspace
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == device) {
                b->dev = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                break;
            }
    }
    b->dev.device = device;
    b->dev.n = at;
    const int rc = upload_tags(*b);
    if (rc != HK_OK) {
        em_close(b);
        return rc;
    }
    *out = b;
    return HK_OK;
}

void em_close(EmBatch *b) {
    if (!b) return;
    DeviceState &d = b->dev;
    const bool any = d.tag_cap || d.arc_cap || d.clip_cap || d.weight_cap ||
                     d.bitmap_cap || d.list_cap || d.merge_cap || d.uploaded;
    if (any) {
        wait_for_uploads(d);
        bool parked = false;
        {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            if (g_pool.size() < POOL_MAX) {
                g_pool.push_back(d);
                parked = true;
            }
        }
        if (!parked) free_workspace(d);
    }
    delete b;
}

int em_own(hk_clip &c, hipStream_t s, EmBatch **out) {
    if (!c.own) {
        hk_clip *one = &c;
        const int rc = em_open(&one, 1, s, &c.own);
        if (rc != HK_OK) return rc;
    }
    c.own->stream = s;
    *out = c.own;
    return HK_OK;
}

void em_release(hk_clip &c) {
    em_close(c.own);
    c.own = nullptr;
}

void em_release_cached() {
    std::vector<DeviceState> all;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        all.swap(g_pool);
    }
    for (DeviceState &d : all) free_workspace(d);
}

int em_estep(EmBatch &b, const std::vector<int> &act, bool fit_too) {
    DeviceState &d = b.dev;
    DeviceGuard guard(b.device);
    const hipStream_t s = b.stream;
    int rc = stage_clips(b, act);
    if (rc != HK_OK) return rc;
    if (b.rows == 0) return HK_OK;

    // windows, radius gates and buffer offsets (host)
    long woff = 0, boff = 0;
    int max_tags = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        hk_clip &c = *b.clips[(size_t)act[j]];
        const int base = b.tag_base[(size_t)act[j]];
        const double rw = c.p.radius_window;
        max_tags = std::max(max_tags, c.n);
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            Arc &arc = c.arcs[k];
            const long row = b.arc_base[j] + (long)k;
            double *p = d.h_arc_params + row * AP;
            long *w = d.h_arc_win + row * AW;
            p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
            p[4] = arc.f_peak; p[7] = 0.0;
            int lo = 0, hi = 0;
            if (!arc.tags.empty()) {
                const double m = median_tag_radius(c, arc.tags);
                p[5] = m * (1 - rw);
                p[6] = m * (1 + rw);
                int f_min, f_max;
                arc_frame_range(c, arc, f_min, f_max);
                if (f_max > f_min) {    // range(f_min, f_max): end exclusive
                    lo = c.start[f_min];
                    hi = c.start[f_max];
                }
            } else {
                p[5] = p[6] = 0.0;
            }
            arc.w_lo = lo;
            arc.w_hi = hi;
            arc.w_off = woff;
            w[0] = base + lo; w[1] = base + hi; w[2] = woff; w[3] = boff;
            woff += hi - lo;
            boff += ((hi - lo) + 63) / 64;
        }
    }
    rc = ensure_weight_capacity(d, woff, boff);
    if (rc != HK_OK) return rc;
    const unsigned A = (unsigned)b.rows;
    const TagArrays tg = tag_arrays(d);
    hipLaunchKernelGGL(estep_raw_k, dim3(A), dim3(256), 0, s, tg, d.arc_params,
                       d.arc_win, d.weights, d.arc_clip, d.clip_tab);
    HK_LAUNCH_CHECK("estep_raw_k");
    if (max_tags > 0) {
        hipLaunchKernelGGL(estep_total_k,
                           dim3((unsigned)((max_tags + 255) / 256),
                                (unsigned)act.size()),
                           dim3(256), 0, s, d.clip_tab, d.arc_win, d.weights,
                           d.total);
        HK_LAUNCH_CHECK("estep_total_k");
    }
    hipLaunchKernelGGL(estep_norm_k, dim3(A), dim3(256), 0, s, tg, d.arc_win,
                       d.weights, d.total, (unsigned long long *)d.bitmap);
    HK_LAUNCH_CHECK("estep_norm_k");
    if (fit_too) {
        // the M-step reads the arc tables as they stand (the E-step does not
        // change the parameters) and the weights where the E-step left them
        rc = launch_moments(b, 0);
        if (rc != HK_OK) return rc;
    }
    { const int src = wait_for_stream(b); if (src != HK_OK) return src; }

    // arc.tags <- membership bits (ascending by construction)
    for (size_t j = 0; j < act.size(); ++j) {
        hk_clip &c = *b.clips[(size_t)act[j]];
        const int base = b.tag_base[(size_t)act[j]];
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            Arc &arc = c.arcs[k];
            arc.tags.clear();
            const long *w = d.h_arc_win + (b.arc_base[j] + (long)k) * AW;
            const long words = ((w[1] - w[0]) + 63) / 64;
            for (long q = 0; q < words; ++q) {
                uint64_t bits = d.h_bitmap[w[3] + q];
                while (bits) {
                    const int bit = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    arc.tags.push_back((int)(w[0] - base + q * 64 + bit));
                }
            }
        }
    }
    if (fit_too) apply_fit(b, act);
    return HK_OK;
}

int em_fit(EmBatch &b, const std::vector<int> &act) {
    DeviceGuard guard(b.device);
    const int rc = run_moments(b, act, 0);
    if (rc != HK_OK) return rc;
    if (b.rows > 0) apply_fit(b, act);
    return HK_OK;
}

int em_tilt(EmBatch &b, const std::vector<int> &act) {
    DeviceGuard guard(b.device);
    const int rc = run_moments(b, act, 1);
    if (rc != HK_OK) return rc;
    for (size_t j = 0; j < act.size(); ++j) {
        hk_clip &c = *b.clips[(size_t)act[j]];
        double sum = 0.0;
        int count = 0;
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            const double *o = b.dev.h_arc_out + (b.arc_base[j] + (long)k) * AP;
            if (o[0] == 0.0) continue;
            const double e_x = o[1], e_y = o[2];
            if (acceleration_ok(c, e_y)) {
                sum += std::atan(e_x / e_y);
                ++count;
            }
        }
        c.tilt = count > 0 ? sum / count : 0.0;
    }
    return HK_OK;
}

int em_member_distances(EmBatch &b, const std::vector<int> &act,
                        std::vector<std::vector<double>> &dist,
                        std::vector<std::vector<long>> &offsets) {
    DeviceState &d = b.dev;
    DeviceGuard guard(b.device);
    dist.assign(act.size(), std::vector<double>());
    offsets.assign(act.size(), std::vector<long>());
    int rc = stage_clips(b, act);
    if (rc != HK_OK) return rc;
    long total = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        const hk_clip &c = *b.clips[(size_t)act[j]];
        std::vector<long> &off = offsets[j];
        off.assign(c.arcs.size() + 1, 0);
        long mine = 0;
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            off[k] = mine;
            mine += (long)c.arcs[k].tags.size();
        }
        off[c.arcs.size()] = mine;
        dist[j].assign((size_t)mine, 0.0);
        total += mine;
    }
    if (b.rows == 0 || total == 0) return HK_OK;
    rc = ensure_list_capacity(d, total);
    if (rc != HK_OK) return rc;
    long pos = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        const hk_clip &c = *b.clips[(size_t)act[j]];
        const int base = b.tag_base[(size_t)act[j]];
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            const Arc &arc = c.arcs[k];
            const long row = b.arc_base[j] + (long)k;
            double *p = d.h_arc_params + row * AP;
            p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
            p[4] = arc.f_peak; p[5] = p[6] = p[7] = 0.0;
            long *w = d.h_arc_win + row * AW;
            w[0] = pos;
            for (int i : arc.tags) d.h_list[pos++] = base + i;
            w[1] = pos; w[2] = -1; w[3] = 0;
        }
    }
    hipLaunchKernelGGL(member_dist_k, dim3((unsigned)b.rows), dim3(64), 0,
                       b.stream, tag_arrays(d), d.arc_params, d.arc_win, d.list,
                       d.list_out, d.arc_clip, d.clip_tab);
    HK_LAUNCH_CHECK("member_dist_k");
    { const int src = wait_for_stream(b); if (src != HK_OK) return src; }
    pos = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        std::copy(d.h_list_out + pos, d.h_list_out + pos + (long)dist[j].size(),
                  dist[j].begin());
        pos += (long)dist[j].size();
    }
    return HK_OK;
}

// All candidate pairs of a merge pass of every clip of `act` in one launch.
int em_merge_trials(EmBatch &b, const std::vector<int> &act,
                    const std::vector<MergeInput> &in,
                    std::vector<std::vector<MergeTrial>> &out) {
    DeviceState &d = b.dev;
    DeviceGuard guard(b.device);
    out.assign(act.size(), std::vector<MergeTrial>());
    int rc = stage_clips(b, act);
    if (rc != HK_OK) return rc;
    size_t npairs = 0, ngood = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        out[j].resize(in[j].pairs.size() / 2);
        npairs += in[j].pairs.size() / 2;
        ngood += in[j].good.size();
    }
    if (npairs == 0) return HK_OK;
    const size_t A = (size_t)b.rows;
    // one staging block: clip table | arc_clip | params | good_off | good |
    // pairs, then the results
    const size_t b_clip = 0;
    const size_t b_aclip = b_clip + sizeof(ClipDev) * act.size();
    size_t b_params = b_aclip + sizeof(int) * A;
    b_params = (b_params + 7) & ~(size_t)7;
    const size_t b_off = b_params + sizeof(double) * 5 * A;
    const size_t b_good = b_off + sizeof(int) * (A + 1);
    const size_t b_pairs = b_good + sizeof(int) * ngood;
    size_t b_out = b_pairs + sizeof(int) * 2 * npairs;
    b_out = (b_out + 15) & ~(size_t)15;
    const size_t total = b_out + sizeof(MergeTrial) * npairs;
    if (total > d.merge_cap) {
        const size_t cap = std::max(total * 2, (size_t)1 << 20);
        (void)hipFree(d.merge_buf);
        (void)hipHostFree(d.h_merge_buf);
        d.merge_buf = nullptr;
        d.h_merge_buf = nullptr;
        d.merge_cap = 0;
        HK_HIP(dev_alloc(d.merge_buf, cap));
        HK_HIP(pin_alloc(d.h_merge_buf, cap));
        d.merge_cap = cap;
    }
    char *h = d.h_merge_buf;
    std::memcpy(h + b_clip, d.h_clip_tab, sizeof(ClipDev) * act.size());
    std::memcpy(h + b_aclip, d.h_arc_clip, sizeof(int) * A);
    double *hp = reinterpret_cast<double *>(h + b_params);
    int *hoff = reinterpret_cast<int *>(h + b_off);
    int *hgood = reinterpret_cast<int *>(h + b_good);
    int *hpairs = reinterpret_cast<int *>(h + b_pairs);
    size_t g_at = 0, p_at = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        const hk_clip &c = *b.clips[(size_t)act[j]];
        const int base = b.tag_base[(size_t)act[j]];
        const int row0 = b.arc_base[j];
        const bool used = !in[j].pairs.empty();
        for (size_t k = 0; k < c.arcs.size(); ++k) {
            const Arc &arc = c.arcs[k];
            double *p = hp + 5 * ((size_t)row0 + k);
            p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
            p[4] = arc.f_peak;
            // (a clip without pairs hands no lists over: empty ranges)
            hoff[(size_t)row0 + k] =
                (int)g_at + (used ? in[j].good_off[k] : 0);
        }
        if (!used) continue;
        for (int i : in[j].good) hgood[g_at++] = base + i;
        for (int r : in[j].pairs) hpairs[p_at++] = row0 + r;
    }
    hoff[A] = (int)g_at;
    HK_HIP(hipMemcpyAsync(d.merge_buf, h, b_out, hipMemcpyHostToDevice,
                          b.stream));
    char *g = d.merge_buf;
    hipLaunchKernelGGL(merge_trials_k, dim3((unsigned)((npairs + 63) / 64)),
                       dim3(64), 0, b.stream, tag_arrays(d),
                       reinterpret_cast<const double *>(g + b_params),
                       reinterpret_cast<const int *>(g + b_off),
                       reinterpret_cast<const int *>(g + b_good),
                       reinterpret_cast<const int *>(g + b_pairs), (int)npairs,
                       reinterpret_cast<MergeTrial *>(h + b_out),
                       reinterpret_cast<const int *>(g + b_aclip),
                       reinterpret_cast<const ClipDev *>(g + b_clip));
    HK_LAUNCH_CHECK("merge_trials_k");
    { const int src = wait_for_stream(b); if (src != HK_OK) return src; }
    const MergeTrial *res = reinterpret_cast<const MergeTrial *>(h + b_out);
    size_t at = 0;
    for (size_t j = 0; j < act.size(); ++j) {
        std::copy(res + at, res + at + out[j].size(), out[j].begin());
        at += out[j].size();
    }
    return HK_OK;
}

}  // namespace hk

extern "C" int hk_clip_get_weights(hk_clip *c, int *tag, int *arc,
                                   double *weight, int cap, void *stream) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_get_weights: null clip");
    if (!c->own)
        return hk::fail(HK_ERR_ARG,
                        "hk_clip_get_weights: run calculate_weights first");
    hk::DeviceState &d = c->own->dev;
    hk::DeviceGuard guard(c->device);
    int n = 0;
    std::vector<double> w;
    for (size_t k = 0; k < c->arcs.size(); ++k) {
        const hk::Arc &a = c->arcs[k];
        if (a.w_off < 0 || a.w_hi <= a.w_lo) continue;
        w.resize((size_t)(a.w_hi - a.w_lo));
        HK_HIP(hipMemcpyAsync(w.data(), d.weights + a.w_off,
                              sizeof(double) * w.size(), hipMemcpyDeviceToHost,
                              (hipStream_t)stream));
        HK_HIP(hipStreamSynchronize((hipStream_t)stream));
        for (size_t j = 0; j < w.size(); ++j) {
            if (!(w[j] > 0.0)) continue;
            if (n < cap) {
                if (tag) tag[n] = a.w_lo + (int)j;
                if (arc) arc[n] = (int)k;
                if (weight) weight[n] = w[j];
            }
            ++n;
        }
    }
    return n;
}
