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
    const long *__restrict__ arc_win, double *__restrict__ weights, double sn,
    double cs, double sigmasq) {
    const int a = blockIdx.x;
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

__global__ __launch_bounds__(256) void estep_total_k(
    int n, int narcs, const long *__restrict__ arc_win,
    const double *__restrict__ weights, double *__restrict__ total) {
    __shared__ long win[256 * 3];
    const int i = blockIdx.x * 256 + threadIdx.x;
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
    const int *__restrict__ list, double *__restrict__ arc_out, double sn,
    double cs, int what) {
    const int a = blockIdx.x;
    const int lane = threadIdx.x;
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
    double *__restrict__ list_out, double sn, double cs) {
    const int a = blockIdx.x;
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
__global__ __launch_bounds__(64) void merge_trials_k(
    TagArrays tg, const double *__restrict__ params5,
    const int *__restrict__ good_off, const int *__restrict__ good,
    const int *__restrict__ pairs, int npairs, hk::MergeTrial *__restrict__ out,
    double sn, double cs, double maxd, int min_tags) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= npairs) return;
    const int i1 = pairs[2 * q], i2 = pairs[2 * q + 1];
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

TagArrays tag_arrays(const hk::DeviceState &d) {
    return TagArrays{d.frame, d.x, d.y, d.r, d.tw};
}

int ensure_arc_capacity(hk::DeviceState &d, int narcs) {
    if (narcs <= d.arc_cap) return HK_OK;
    const int cap = std::max(256, narcs * 2);
    (void)hipFree(d.arc_params); (void)hipFree(d.arc_win);
    (void)hipFree(d.arc_out);
    (void)hipHostFree(d.h_arc_params); (void)hipHostFree(d.h_arc_win);
    (void)hipHostFree(d.h_arc_out);
    d.arc_params = nullptr; d.arc_win = nullptr; d.arc_out = nullptr;
    d.h_arc_params = nullptr; d.h_arc_win = nullptr; d.h_arc_out = nullptr;
    d.arc_cap = 0;
    HK_HIP(dev_alloc(d.arc_params, (size_t)cap * AP));
    HK_HIP(dev_alloc(d.arc_win, (size_t)cap * AW));
    HK_HIP(dev_alloc(d.arc_out, (size_t)cap * AP));
    HK_HIP(pin_alloc(d.h_arc_params, (size_t)cap * AP));
    HK_HIP(pin_alloc(d.h_arc_win, (size_t)cap * AW));
    HK_HIP(pin_alloc(d.h_arc_out, (size_t)cap * AP));
    d.arc_cap = cap;
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
        (void)hipFree(d.bitmap);
        (void)hipHostFree(d.h_bitmap);
        d.bitmap = nullptr;
        d.h_bitmap = nullptr;
        d.bitmap_cap = 0;
        HK_HIP(dev_alloc(d.bitmap, (size_t)cap));
        HK_HIP(pin_alloc(d.h_bitmap, (size_t)cap));
        d.bitmap_cap = cap;
    }
    return HK_OK;
}

int ensure_list_capacity(hk::DeviceState &d, long entries) {
    if (entries <= d.list_cap) return HK_OK;
    const long cap = std::max(entries * 2, 1L << 14);
    (void)hipFree(d.list); (void)hipFree(d.list_out);
    (void)hipHostFree(d.h_list); (void)hipHostFree(d.h_list_out);
    d.list = nullptr; d.list_out = nullptr;
    d.h_list = nullptr; d.h_list_out = nullptr;
    d.list_cap = 0;
    HK_HIP(dev_alloc(d.list, (size_t)cap));
    HK_HIP(dev_alloc(d.list_out, (size_t)cap));
    HK_HIP(pin_alloc(d.h_list, (size_t)cap));
    HK_HIP(pin_alloc(d.h_list_out, (size_t)cap));
    d.list_cap = cap;
    return HK_OK;
}

// Fills the pinned arc tables.  Arcs with dense weights keep their window;
// the others get their explicit member list appended to h_list.
int stage_arcs(hk_clip &c, long &list_entries) {
    hk::DeviceState &d = c.dev;
    const int A = (int)c.arcs.size();
    int rc = ensure_arc_capacity(d, A);
    if (rc != HK_OK) return rc;
    long total = 0;
    for (const hk::Arc &arc : c.arcs)
        if (arc.w_off < 0) total += (long)arc.tags.size();
    rc = ensure_list_capacity(d, total);
    if (rc != HK_OK) return rc;
    long pos = 0;
    for (int k = 0; k < A; ++k) {
        const hk::Arc &arc = c.arcs[k];
        double *p = d.h_arc_params + (long)k * AP;
        p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
        p[4] = arc.f_peak; p[5] = p[6] = p[7] = 0.0;
        long *w = d.h_arc_win + (long)k * AW;
        if (arc.w_off >= 0) {
            w[0] = arc.w_lo; w[1] = arc.w_hi; w[2] = arc.w_off; w[3] = 0;
        } else {
            w[0] = pos;
            for (int i : arc.tags) d.h_list[pos++] = i;
            w[1] = pos; w[2] = -1; w[3] = 0;
        }
    }
    list_entries = pos;
    return HK_OK;
}

int upload_arcs(hk_clip &c, long list_entries, hipStream_t s) {
    hk::DeviceState &d = c.dev;
    const size_t A = c.arcs.size();
    if (A == 0) return HK_OK;
    HK_HIP(hipMemcpyAsync(d.arc_params, d.h_arc_params, sizeof(double) * A * AP,
                          hipMemcpyHostToDevice, s));
    HK_HIP(hipMemcpyAsync(d.arc_win, d.h_arc_win, sizeof(long) * A * AW,
                          hipMemcpyHostToDevice, s));
    if (list_entries > 0)
        HK_HIP(hipMemcpyAsync(d.list, d.h_list, sizeof(int) * list_entries,
                              hipMemcpyHostToDevice, s));
    return HK_OK;
}

int run_moments(hk_clip &c, hipStream_t s, int what) {
    hk::DeviceState &d = c.dev;
    const int A = (int)c.arcs.size();
    if (A == 0) return HK_OK;
    long entries = 0;
    int rc = stage_arcs(c, entries);
    if (rc != HK_OK) return rc;
    rc = upload_arcs(c, entries, s);
    if (rc != HK_OK) return rc;
    const hk::Rot rot = hk::rotation(c);
    hipLaunchKernelGGL(arc_moments_k, dim3(A), dim3(64), 0, s, tag_arrays(d),
                       d.arc_params, d.arc_win, d.weights, d.list, d.arc_out,
                       rot.s, rot.c, what);
    HK_LAUNCH_CHECK("arc_moments_k");
    HK_HIP(hipMemcpyAsync(d.h_arc_out, d.arc_out, sizeof(double) * A * AP,
                          hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));
    return HK_OK;
}

}  // namespace

namespace hk {

// Device workspaces outlive the clips that use them: hipMalloc / hipFree
// (the latter synchronises the whole device) cost more than a whole EM pass
// on a clip, so a released workspace is parked here, per device, and the next
// clip on that device takes it over.  hk_release_cached_memory() frees them.
// As many are kept as clips can be in steps 3-4 at a time in a pipelined batch
// (one per builder thread, up to ~30 with more cores per GPU): with fewer, a
// batch frees and allocates workspaces all the time, and every hipFree holds
// up the detection thread's launches for milliseconds.  A workspace of a
// 50 000-tag clip is a few MB.
namespace {
constexpr size_t POOL_MAX = 48;
std::mutex g_pool_mutex;
std::vector<DeviceState> g_pool;

// The tag uploads of em_prepare() are asynchronous copies out of the
// workspace's pinned block.  A clip without arcs never synchronises its
// stream (every EM call returns early), and an error path may skip the wait
// too: before the workspace is parked, freed or its block refilled, the copies
// must have left it -- otherwise the next owner, on another stream, could
// overwrite h_tags under them or see its own uploads overtaken by them.
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
    (void)hipFree(d.frame); (void)hipFree(d.x); (void)hipFree(d.y);
    (void)hipFree(d.r); (void)hipFree(d.tw); (void)hipFree(d.total);
    (void)hipHostFree(d.h_tags);
    (void)hipFree(d.arc_params); (void)hipFree(d.arc_win);
    (void)hipFree(d.arc_out); (void)hipFree(d.weights);
    (void)hipFree(d.bitmap); (void)hipFree(d.list); (void)hipFree(d.list_out);
    (void)hipHostFree(d.h_arc_params); (void)hipHostFree(d.h_arc_win);
    (void)hipHostFree(d.h_arc_out); (void)hipHostFree(d.h_bitmap);
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
}  // namespace

int em_prepare(hk_clip &c, hipStream_t s) {
    DeviceState &d = c.dev;
    if (d.ready) return HK_OK;
    DeviceGuard guard(c.device);
    if (!guard.ok)
        return fail(HK_ERR_HIP, "cannot select device %d", c.device);
    if (!d.tag_cap && !d.arc_cap) {         // take over a parked workspace
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == c.device) {
                d = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                break;
            }
    }
    d.device = c.device;
    d.n = c.n;
    const int rc = ensure_tag_capacity(d, c.n);
    if (rc != HK_OK) return rc;
    const size_t n = (size_t)c.n;
    wait_for_uploads(d);        // (a workspace that em_release parked has none)
    if (n) {
        // Through the workspace's pinned block: a copy from the clip's
        // pageable vectors is a synchronous call of ~0.2 ms each that holds
        // up other threads' launches (seen in a HIP API trace of a batch).
        // The block is the workspace's own; an event behind the copies is
        // waited for before the workspace is handed on (wait_for_uploads).
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
        std::memcpy(hx, c.x.data(), sizeof(double) * n);
        std::memcpy(hy, c.y.data(), sizeof(double) * n);
        std::memcpy(hr, c.r.data(), sizeof(double) * n);
        std::memcpy(htw, c.tw.data(), sizeof(double) * n);
        std::memcpy(hf, c.frame.data(), sizeof(int) * n);
        HK_HIP(hipMemcpyAsync(d.frame, hf, sizeof(int) * n,
                              hipMemcpyHostToDevice, s));
        HK_HIP(hipMemcpyAsync(d.x, hx, sizeof(double) * n,
                              hipMemcpyHostToDevice, s));
        HK_HIP(hipMemcpyAsync(d.y, hy, sizeof(double) * n,
                              hipMemcpyHostToDevice, s));
        HK_HIP(hipMemcpyAsync(d.r, hr, sizeof(double) * n,
                              hipMemcpyHostToDevice, s));
        // (pending from the first copy on, so that an error half-way still
        // waits: the event then completes at once or behind what was queued)
        d.upload_pending = true;
        hipError_t e = hipMemcpyAsync(d.tw, htw, sizeof(double) * n,
                                      hipMemcpyHostToDevice, s);
        const hipError_t e2 = hipEventRecord((hipEvent_t)d.uploaded, s);
        if (e == hipSuccess) e = e2;
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(s);
            d.upload_pending = false;
            HK_HIP(e);
        }
    }
    d.ready = true;
    return HK_OK;
}

void em_release(hk_clip &c) {
    DeviceState &d = c.dev;
    if (!d.tag_cap && !d.arc_cap && !d.weight_cap && !d.bitmap_cap &&
        !d.list_cap && !d.merge_cap)
        return;
    d.ready = false;
    wait_for_uploads(d);
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (g_pool.size() < POOL_MAX) {
            g_pool.push_back(d);
            d = DeviceState();
            return;
        }
    }
    free_workspace(d);
}

void em_release_cached() {
    std::vector<DeviceState> all;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        all.swap(g_pool);
    }
    for (DeviceState &d : all) free_workspace(d);
}

int em_estep(hk_clip &c, hipStream_t s) {
    DeviceState &d = c.dev;
    DeviceGuard guard(c.device);
    const int A = (int)c.arcs.size();
    if (A == 0) return HK_OK;
    int rc = ensure_arc_capacity(d, A);
    if (rc != HK_OK) return rc;

    // windows, radius gates and buffer offsets (host)
    long woff = 0, boff = 0;
    const double rw = c.p.radius_window;
    for (int k = 0; k < A; ++k) {
        Arc &arc = c.arcs[k];
        double *p = d.h_arc_params + (long)k * AP;
        long *w = d.h_arc_win + (long)k * AW;
        p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
        p[4] = arc.f_peak; p[7] = 0.0;
        int lo = 0, hi = 0;
        if (!arc.tags.empty()) {
            const double m = median_tag_radius(c, arc.tags);
            p[5] = m * (1 - rw);
            p[6] = m * (1 + rw);
            int f_min, f_max;
            arc_frame_range(c, arc, f_min, f_max);
            if (f_max > f_min) {        // range(f_min, f_max): end exclusive
                lo = c.start[f_min];
                hi = c.start[f_max];
            }
        } else {
            p[5] = p[6] = 0.0;
        }
        arc.w_lo = lo;
        arc.w_hi = hi;
        arc.w_off = woff;
        w[0] = lo; w[1] = hi; w[2] = woff; w[3] = boff;
        woff += hi - lo;
        boff += ((hi - lo) + 63) / 64;
    }
    rc = ensure_weight_capacity(d, woff, boff);
    if (rc != HK_OK) return rc;
    rc = upload_arcs(c, 0, s);
    if (rc != HK_OK) return rc;

    const Rot rot = rotation(c);
    const TagArrays tg = tag_arrays(d);
    hipLaunchKernelGGL(estep_raw_k, dim3(A), dim3(256), 0, s, tg, d.arc_params,
                       d.arc_win, d.weights, rot.s, rot.c, c.p.sigmasq);
    HK_LAUNCH_CHECK("estep_raw_k");
    if (c.n > 0) {
        hipLaunchKernelGGL(estep_total_k, dim3((c.n + 255) / 256), dim3(256),
                           0, s, c.n, A, d.arc_win, d.weights, d.total);
        HK_LAUNCH_CHECK("estep_total_k");
    }
    hipLaunchKernelGGL(estep_norm_k, dim3(A), dim3(256), 0, s, tg, d.arc_win,
                       d.weights, d.total,
                       (unsigned long long *)d.bitmap);
    HK_LAUNCH_CHECK("estep_norm_k");
    if (boff > 0)
        HK_HIP(hipMemcpyAsync(d.h_bitmap, d.bitmap, sizeof(uint64_t) * boff,
                              hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));

    // arc.tags <- membership bits (ascending by construction)
    for (int k = 0; k < A; ++k) {
        Arc &arc = c.arcs[k];
        arc.tags.clear();
        const long *w = d.h_arc_win + (long)k * AW;
        const long words = ((w[1] - w[0]) + 63) / 64;
        for (long j = 0; j < words; ++j) {
            uint64_t bits = d.h_bitmap[w[3] + j];
            while (bits) {
                const int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                arc.tags.push_back((int)(w[0] + j * 64 + b));
            }
        }
    }
    return HK_OK;
}

int em_fit(hk_clip &c, hipStream_t s) {
    DeviceGuard guard(c.device);
    const int rc = run_moments(c, s, 0);
    if (rc != HK_OK) return rc;
    for (size_t k = 0; k < c.arcs.size(); ++k) {
        const double *o = c.dev.h_arc_out + (long)k * AP;
        if (o[0] == 0.0) continue;
        Arc &arc = c.arcs[k];
        arc.a = o[1]; arc.b = o[2]; arc.c = o[3]; arc.e = o[4];
        arc.f_peak = o[5];
    }
    return HK_OK;
}

int em_tilt(hk_clip &c, hipStream_t s) {
    DeviceGuard guard(c.device);
    const int rc = run_moments(c, s, 1);
    if (rc != HK_OK) return rc;
    double sum = 0.0;
    int count = 0;
    for (size_t k = 0; k < c.arcs.size(); ++k) {
        const double *o = c.dev.h_arc_out + (long)k * AP;
        if (o[0] == 0.0) continue;
        const double e_x = o[1], e_y = o[2];
        if (acceleration_ok(c, e_y)) {
            sum += std::atan(e_x / e_y);
            ++count;
        }
    }
    c.tilt = count > 0 ? sum / count : 0.0;
    return HK_OK;
}

int em_member_distances(hk_clip &c, hipStream_t s, std::vector<double> &dist,
                        std::vector<long> &offsets) {
    DeviceState &d = c.dev;
    DeviceGuard guard(c.device);
    const int A = (int)c.arcs.size();
    offsets.assign((size_t)A + 1, 0);
    long total = 0;
    for (int k = 0; k < A; ++k) {
        offsets[k] = total;
        total += (long)c.arcs[k].tags.size();
    }
    offsets[A] = total;
    dist.assign((size_t)total, 0.0);
    if (A == 0 || total == 0) return HK_OK;
    int rc = ensure_arc_capacity(d, A);
    if (rc != HK_OK) return rc;
    rc = ensure_list_capacity(d, total);
    if (rc != HK_OK) return rc;
    for (int k = 0; k < A; ++k) {
        const Arc &arc = c.arcs[k];
        double *p = d.h_arc_params + (long)k * AP;
        p[0] = arc.a; p[1] = arc.b; p[2] = arc.c; p[3] = arc.e;
        p[4] = arc.f_peak; p[5] = p[6] = p[7] = 0.0;
        long *w = d.h_arc_win + (long)k * AW;
        w[0] = offsets[k]; w[1] = offsets[k + 1]; w[2] = -1; w[3] = 0;
        std::copy(arc.tags.begin(), arc.tags.end(), d.h_list + offsets[k]);
    }
    rc = upload_arcs(c, total, s);
    if (rc != HK_OK) return rc;
    const Rot rot = rotation(c);
    hipLaunchKernelGGL(member_dist_k, dim3(A), dim3(64), 0, s, tag_arrays(d),
                       d.arc_params, d.arc_win, d.list, d.list_out, rot.s,
                       rot.c);
    HK_LAUNCH_CHECK("member_dist_k");
    HK_HIP(hipMemcpyAsync(d.h_list_out, d.list_out, sizeof(double) * total,
                          hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));
    std::copy(d.h_list_out, d.h_list_out + total, dist.begin());
    return HK_OK;
}

// All candidate pairs of a merge pass in one launch.  good_off/good: the
// arcs' good-tag lists (CSR over the current arc list); pairs: (i1, i2)
// positions in that list.
int em_merge_trials(hk_clip &c, hipStream_t s, const std::vector<int> &good_off,
                    const std::vector<int> &good, const std::vector<int> &pairs,
                    std::vector<MergeTrial> &out) {
    DeviceState &d = c.dev;
    DeviceGuard guard(c.device);
    const size_t A = c.arcs.size();
    const size_t npairs = pairs.size() / 2;
    out.resize(npairs);
    if (npairs == 0) return HK_OK;
    // one staging block: params | good_off | good | pairs, then the results
    const size_t b_params = 0;
    const size_t b_off = b_params + sizeof(double) * 5 * A;
    const size_t b_good = b_off + sizeof(int) * (A + 1);
    const size_t b_pairs = b_good + sizeof(int) * good.size();
    size_t b_out = b_pairs + sizeof(int) * pairs.size();
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
    double *hp = reinterpret_cast<double *>(h + b_params);
    for (size_t k = 0; k < A; ++k) {
        const Arc &arc = c.arcs[k];
        hp[5 * k] = arc.a;
        hp[5 * k + 1] = arc.b;
        hp[5 * k + 2] = arc.c;
        hp[5 * k + 3] = arc.e;
        hp[5 * k + 4] = arc.f_peak;
    }
    std::memcpy(h + b_off, good_off.data(), sizeof(int) * (A + 1));
    std::memcpy(h + b_good, good.data(), sizeof(int) * good.size());
    std::memcpy(h + b_pairs, pairs.data(), sizeof(int) * pairs.size());
    HK_HIP(hipMemcpyAsync(d.merge_buf, h, b_out, hipMemcpyHostToDevice, s));
    const Rot rot = rotation(c);
    char *g = d.merge_buf;
    hipLaunchKernelGGL(merge_trials_k, dim3((unsigned)((npairs + 63) / 64)),
                       dim3(64), 0, s, tag_arrays(d),
                       reinterpret_cast<const double *>(g + b_params),
                       reinterpret_cast<const int *>(g + b_off),
                       reinterpret_cast<const int *>(g + b_good),
                       reinterpret_cast<const int *>(g + b_pairs), (int)npairs,
                       reinterpret_cast<MergeTrial *>(g + b_out), rot.s, rot.c,
                       c.p.max_distance_pixels, c.p.min_tags_per_arc);
    HK_LAUNCH_CHECK("merge_trials_k");
    HK_HIP(hipMemcpyAsync(h + b_out, g + b_out, sizeof(MergeTrial) * npairs,
                          hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));
    std::memcpy(out.data(), h + b_out, sizeof(MergeTrial) * npairs);
    return HK_OK;
}

}  // namespace hk

extern "C" int hk_clip_get_weights(hk_clip *c, int *tag, int *arc,
                                   double *weight, int cap, void *stream) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_get_weights: null clip");
    hk::DeviceState &d = c->dev;
    if (!d.ready)
        return hk::fail(HK_ERR_ARG,
                        "hk_clip_get_weights: run calculate_weights first");
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
