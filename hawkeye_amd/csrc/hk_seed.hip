// K9/K10 -- step 3 (preliminary arcs) on the device.
//
// Replaces the per-seed work of construct_arcs (reference tracker.py:487-652)
// and build_neighbor_lists (tracker.py:654-696).  The reference visits the
// seed tags one by one; a seed either fails or -- rarely -- yields an arc that
// takes its member tags out of play.  hk_arcs.cpp keeps that order exactly by
// speculation: seeds are evaluated under the assumption that every earlier
// seed still uncommitted fails, and a committer, in seed order, re-runs those
// whose assumption was broken (a tag they looked at was taken by an arc
// accepted in the meantime).  Here the evaluation side of that scheme runs on
// the GPU: one wave per seed, a window of seeds per launch, every seed of the
// window against the same snapshot of the "taken" flags; the host commits the
// window in order and re-runs the few stale seeds itself.
//
//   seed_neighbors_k  one wave per tag: candidates of the frames within
//                     max_frame_gap_in_arc + 1, speed cap, ordered by (speed,
//                     tag index) by rank counting in LDS.
//   seed_run_k        one wave per seed: the growth loop of a trial arc with
//                     its member and frontier lists in LDS.  Lanes share the
//                     data-parallel parts (distances to the frontier, the
//                     check of all members after a refit, list scans); the
//                     moment sums of a refit run in ascending member order on
//                     11 lanes, one per sum, so that they equal the host's
//                     sequential sums bit for bit; the solve is hk_fit.h,
//                     compiled for both sides.  No FMA contraction, IEEE fp64
//                     divide and sqrt: a device result equals the host's
//                     run_seed() result exactly (hk_set_self_check(1) compares
//                     every one).
//
// Anything the fixed-size lists cannot hold (very long arcs, very dense
// clutter, non-finite distances) is flagged and left to the host code.

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "hk_seed.h"

// Build option: the product library (`make`, i.e. DEVICE_SEEDS=0) leaves the
// device side of step 3 out: it is opt-in at run time anyway, HK_STEP3_DEVICE,
// and the multi-thread host builder is the faster one on every clip measured
// so far.  seed_open() then reports "does not apply here" and hk_arcs.cpp
// takes its host path, as it does for a clip the fixed-size device lists
// cannot hold.  `make DEVICE_SEEDS=1` builds libhawkeye_hip_device_seeds.so
// with it (select with HK_NATIVE_LIB; tests/test_gpu_step3.py).
#ifdef HK_NO_DEVICE_SEEDS
extern "C" int hk_step3_device_built(void) { return 0; }
namespace hk {
struct SeedDevice {};
void seed_release_cached() {}
void seed_close(SeedDevice *) {}
int seed_open(hk_clip &, const SeedParams &, const std::vector<int> &,
              const std::vector<int> &, int, SeedDevice **out) {
    *out = nullptr;
    return HK_OK;
}
int seed_windows(const SeedDevice *) { return 0; }
int seed_round(SeedDevice *, int, const unsigned char *, SeedRound &) {
    return HK_ERR_INTERNAL;
}
}  // namespace hk
#else
extern "C" int hk_step3_device_built(void) { return 1; }

namespace {

using hk::SeedParams;

constexpr int NB_CAP = hk::S3_NB_CAP;
constexpr int MEM_CAP = hk::S3_MEMBERS;
constexpr int FR_CAP = 192;             // frontier entries
constexpr int CONS_CAP = hk::S3_CONSULTED;
constexpr int TAKEN_HALF = 512;         // frames either side of the seed's
constexpr int TAKEN_WORDS = 2 * TAKEN_HALF / 32;

struct Tags {
    const int *frame;
    const double *x, *y;
    const int *start;       // [F + 1]
    int n, F;
};

// ---------------------------------------------------------------------------
// K9: neighbour lists
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void seed_neighbors_k(
    const Tags tg, int gap, int has_vmax, double v_max, int *__restrict__ nb,
    int *__restrict__ nb_len, int *__restrict__ overflow) {
    __shared__ double cv[NB_CAP];
    __shared__ int cj[NB_CAP];
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const int f = tg.frame[i];
    const int lo = f - gap - 1 > 0 ? f - gap - 1 : 0;
    const int hi = f + gap + 2 < tg.F ? f + gap + 2 : tg.F;
    const int j0 = tg.start[lo], j1 = tg.start[hi];
    const double xi = tg.x[i], yi = tg.y[i];
    int m = 0;
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        bool keep = false;
        double v = 0;
        if (j < j1) {
            const int fj = tg.frame[j];
            if (fj != f) {
                const double dx = tg.x[j] - xi, dy = tg.y[j] - yi;
                const int df = fj > f ? fj - f : f - fj;
                v = sqrt(dx * dx + dy * dy) / df;
                keep = has_vmax ? (v <= v_max) : true;
            }
        }
        const unsigned long long mask = __ballot(keep);
        const int at = m + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep && at < NB_CAP) {
            cv[at] = v;
            cj[at] = j;
        }
        m += __popcll(mask);
    }
    if (m > NB_CAP) {
        if (lane == 0) {
            atomicExch(overflow, 1);
            nb_len[i] = 0;
        }
        return;
    }
    __syncthreads();
    // position of each entry when ordered by (speed, tag index)
    for (int e = lane; e < m; e += 64) {
        const double v = cv[e];
        const int j = cj[e];
        int rank = 0;
        for (int k = 0; k < m; ++k) {
            const double u = cv[k];
            rank += (u < v) || (u == v && cj[k] < j);
        }
        nb[(long)i * NB_CAP + rank] = j;
    }
    if (lane == 0) nb_len[i] = m;
}

// ---------------------------------------------------------------------------
// K10: one seed per wave
// ---------------------------------------------------------------------------
struct SeedIO {
    const int *queue, *qpos;
    const unsigned char *done;
    const int *nb, *nb_len;
    int p0, count;
    int *status;            // [count]
    int *n_cons;            // [count]
    int *cons;              // [count][CONS_CAP]
    unsigned *made_count;
    int *made_seed;         // [S3_MADE_CAP] position in the window
    double *made_params;    // [S3_MADE_CAP][5]
    int *made_n;            // [S3_MADE_CAP]
    int *made_members;      // [S3_MADE_CAP][MEM_CAP]
};

struct WaveState {
    int mem_j[MEM_CAP];
    double mem_f[MEM_CAP], mem_x[MEM_CAP], mem_y[MEM_CAP];
    int fr_j[FR_CAP], fr_f[FR_CAP];
    double fr_x[FR_CAP], fr_y[FR_CAP];
    int cons[CONS_CAP];
    unsigned taken[TAKEN_WORDS];
    int n_mem, n_fr, n_cons, fbase, give_up;
};
// give_up values
constexpr int GU_BUDGET = 2;

__device__ __forceinline__ bool tag_done(const SeedIO &io, int j, int pos) {
    return io.qpos[j] < pos || io.done[j] != 0;
}

// Lanes of `mask` whose value v is in list[0 .. n) (n <= 64 * CHUNKS): the
// list sits in registers, 64 entries per chunk, and each value in question is
// broadcast and compared by all lanes at once.
template <int CHUNKS>
__device__ __forceinline__ unsigned long long lanes_in_list(
    unsigned long long mask, int v, const int *list, int n) {
    const int lane = threadIdx.x;
    int held[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c)
        held[c] = c * 64 + lane < n ? list[c * 64 + lane] : -1;
    unsigned long long found = 0ull;
    while (mask) {
        const int k = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int vk = __shfl(v, k, 64);
        bool hit = false;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) hit = hit || held[c] == vk;
        if (__ballot(hit) != 0ull) found |= 1ull << k;
    }
    return found;
}

// Lists the not-yet-done tags of `list` as consulted and puts those that are
// neither in a taken frame nor in the frontier already into the frontier
// (ArcBuilder::add_to_frontier).
__device__ void add_to_frontier(WaveState &w, const Tags &tg, const SeedIO &io,
                                const int *list, int len, int pos) {
    const int lane = threadIdx.x;
    for (int base = 0; base < len; base += 64) {
        const int k = base + lane;
        const int j = k < len ? list[k] : -1;
        // everything about tag j in one round trip
        int qp = 0, fj = 0;
        unsigned char dn = 1;
        double xj = 0, yj = 0;
        if (j >= 0) {
            qp = io.qpos[j];
            dn = io.done[j];
            fj = tg.frame[j];
            xj = tg.x[j];
            yj = tg.y[j];
        }
        const bool live = j >= 0 && !(qp < pos || dn != 0);
        const unsigned long long live_mask = __ballot(live);
        // consulted: append if new
        const int nc = w.n_cons;
        {
            const unsigned long long known =
                lanes_in_list<CONS_CAP / 64>(live_mask, j, w.cons, nc);
            const bool fresh = live && !((known >> lane) & 1ull);
            const unsigned long long mask = __ballot(fresh);
            const int at = nc + __popcll(mask & ((1ull << lane) - 1ull));
            __syncthreads();
            if (fresh) {
                if (at < CONS_CAP) w.cons[at] = j;
                else w.give_up = 1;
            }
            if (lane == 0) {
                const int total = nc + __popcll(mask);
                w.n_cons = total < CONS_CAP ? total : CONS_CAP;
            }
        }
        // frontier
        bool add = live;
        if (live) {
            const int rel = fj - w.fbase;
            if (rel < 0 || rel >= 2 * TAKEN_HALF) {
                w.give_up = 1;
                add = false;
            } else if ((w.taken[rel >> 5] >> (rel & 31)) & 1u) {
                add = false;
            }
        }
        const int nf = w.n_fr;
        const unsigned long long there =
            lanes_in_list<FR_CAP / 64>(__ballot(add), j, w.fr_j, nf);
        add = add && !((there >> lane) & 1ull);
        const unsigned long long mask = __ballot(add);
        const int at = nf + __popcll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (add) {
            if (at < FR_CAP) {
                w.fr_j[at] = j;
                w.fr_f[at] = fj;
                w.fr_x[at] = xj;
                w.fr_y[at] = yj;
            } else {
                w.give_up = 1;
            }
        }
        if (lane == 0) {
            const int total = nf + __popcll(mask);
            w.n_fr = total < FR_CAP ? total : FR_CAP;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void consult_one(WaveState &w, int j) {
    // wave-uniform j
    const int lane = threadIdx.x;
    bool found = false;
    for (int q = lane; q < w.n_cons; q += 64) found = found || w.cons[q] == j;
    const bool any = __ballot(found) != 0ull;
    __syncthreads();
    if (!any && lane == 0) {
        if (w.n_cons < CONS_CAP) w.cons[w.n_cons++] = j;
        else w.give_up = 1;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void seed_run_k(const Tags tg,
                                                 const SeedParams prm,
                                                 const SeedIO io) {
    __shared__ WaveState w;
    const int lane = threadIdx.x;
    const int slot = blockIdx.x;
    const int pos = io.p0 + slot;
    const int t1 = io.queue[pos];
    if (io.done[t1]) {
        if (lane == 0) {
            io.status[slot] = hk::S3_SKIPPED;
            io.n_cons[slot] = 0;
        }
        return;
    }
    const int f1 = tg.frame[t1];
    const double x1 = tg.x[t1], y1 = tg.y[t1];
    if (lane == 0) {
        w.n_cons = 0;
        w.give_up = 0;
        w.fbase = f1 - TAKEN_HALF;
    }
    __syncthreads();

    const int *nb1 = io.nb + (long)t1 * NB_CAP;
    const int len1 = io.nb_len[t1];
    int made = 0;
    int steps = 0;          // growth steps of this seed, all trials
    double p[5] = {0, 0, 0, 0, 0};

    for (int k2 = 0; k2 < len1 && !made; ++k2) {
        const int t2 = nb1[k2];
        if (tag_done(io, t2, pos)) continue;
        consult_one(w, t2);
        const int f2 = tg.frame[t2];
        const double x2 = tg.x[t2], y2 = tg.y[t2];
        // members in ascending tag order, nothing taken but the two frames
        if (lane < TAKEN_WORDS) w.taken[lane] = 0u;
        __syncthreads();
        if (lane == 0) {
            const int a = t1 < t2 ? 0 : 1;          // slot of t1
            w.mem_j[a] = t1;
            w.mem_f[a] = (double)f1;
            w.mem_x[a] = x1;
            w.mem_y[a] = y1;
            w.mem_j[1 - a] = t2;
            w.mem_f[1 - a] = (double)f2;
            w.mem_x[1 - a] = x2;
            w.mem_y[1 - a] = y2;
            w.n_mem = 2;
            w.n_fr = 0;
            const int r1 = f1 - w.fbase, r2 = f2 - w.fbase;
            w.taken[r1 >> 5] |= 1u << (r1 & 31);
            if (r2 >= 0 && r2 < 2 * TAKEN_HALF)
                w.taken[r2 >> 5] |= 1u << (r2 & 31);
            else
                w.give_up = 1;
        }
        __syncthreads();
        add_to_frontier(w, tg, io, nb1, len1, pos);
        add_to_frontier(w, tg, io, io.nb + (long)t2 * NB_CAP, io.nb_len[t2],
                        pos);

        // the parabola through the two tags with the expected acceleration
        p[3] = prm.e0;
        p[4] = (y1 - y2) / (2.0 * p[3] * (f2 - f1)) + 0.5 * (f1 + f2);
        const double dt1 = f1 - p[4];
        p[2] = y1 - p[3] * (dt1 * dt1);
        p[1] = (x1 - x2) / (f1 - f2);
        p[0] = x1 - p[1] * (f1 - p[4]);

        bool bad = false;
        while (w.n_fr > 0 && !w.give_up) {
            // a seed that keeps growing long trial arcs is cheaper on a host
            // core (one wave is slow at this sequential work)
            if (++steps > prm.budget) {
                if (lane == 0) w.give_up = GU_BUDGET;
                __syncthreads();
                break;
            }
            // nearest frontier tag
            double best_d = 0;
            int best = -1, best_at = -1;
            bool nan_seen = false;
            for (int q = lane; q < w.n_fr; q += 64) {
                const double d = sqrt(hk::arc_dist_sq5(
                    p, (double)w.fr_f[q], w.fr_x[q], w.fr_y[q], prm.rot_s,
                    prm.rot_c));
                nan_seen = nan_seen || d != d;
                const int j = w.fr_j[q];
                if (best < 0 || d < best_d || (d == best_d && j < best)) {
                    best_d = d;
                    best = j;
                    best_at = q;
                }
            }
            if (__ballot(nan_seen) != 0ull) {
                if (lane == 0) w.give_up = 1;
                __syncthreads();
                break;
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                const double od = __shfl_xor(best_d, s, 64);
                const int oj = __shfl_xor(best, s, 64);
                const int oa = __shfl_xor(best_at, s, 64);
                if (oj >= 0 && (best < 0 || od < best_d ||
                                (od == best_d && oj < best))) {
                    best_d = od;
                    best = oj;
                    best_at = oa;
                }
            }
            if (best_d > prm.maxd) break;

            // insert into the members, keeping ascending tag order
            const int nm = w.n_mem;
            if (nm >= MEM_CAP) {
                if (lane == 0) w.give_up = 1;
                __syncthreads();
                break;
            }
            const int bf = w.fr_f[best_at];
            const double bx = w.fr_x[best_at], by = w.fr_y[best_at];
            int mv_j[MEM_CAP / 64];
            double mv_f[MEM_CAP / 64], mv_x[MEM_CAP / 64], mv_y[MEM_CAP / 64];
#pragma unroll
            for (int c = 0; c < MEM_CAP / 64; ++c) {
                const int q = c * 64 + lane;
                mv_j[c] = -1;
                if (q < nm && w.mem_j[q] > best) {
                    mv_j[c] = w.mem_j[q];
                    mv_f[c] = w.mem_f[q];
                    mv_x[c] = w.mem_x[q];
                    mv_y[c] = w.mem_y[q];
                }
            }
            int below = 0;
            for (int q = lane; q < nm; q += 64) below += w.mem_j[q] < best;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) below += __shfl_xor(below, s, 64);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < MEM_CAP / 64; ++c) {
                const int q = c * 64 + lane;
                if (mv_j[c] >= 0) {
                    w.mem_j[q + 1] = mv_j[c];
                    w.mem_f[q + 1] = mv_f[c];
                    w.mem_x[q + 1] = mv_x[c];
                    w.mem_y[q + 1] = mv_y[c];
                }
            }
            if (lane == 0) {
                w.mem_j[below] = best;
                w.mem_f[below] = (double)bf;
                w.mem_x[below] = bx;
                w.mem_y[below] = by;
                w.n_mem = nm + 1;
                const int rel = bf - w.fbase;       // in range: it was added
                w.taken[rel >> 5] |= 1u << (rel & 31);
            }
            __syncthreads();
            add_to_frontier(w, tg, io, io.nb + (long)best * NB_CAP,
                            io.nb_len[best], pos);
            // drop the frontier tags of the frame just taken
            {
                const int nf = w.n_fr;
                int kj[FR_CAP / 64], kf[FR_CAP / 64];
                double kx[FR_CAP / 64], ky[FR_CAP / 64];
                int at[FR_CAP / 64];
                int kept = 0;
#pragma unroll
                for (int c = 0; c < FR_CAP / 64; ++c) {
                    const int q = c * 64 + lane;
                    const bool keep = q < nf && w.fr_f[q] != bf;
                    kj[c] = -1;
                    if (keep) {
                        kj[c] = w.fr_j[q];
                        kf[c] = w.fr_f[q];
                        kx[c] = w.fr_x[q];
                        ky[c] = w.fr_y[q];
                    }
                    const unsigned long long mask = __ballot(keep);
                    at[c] = kept + __popcll(mask & ((1ull << lane) - 1ull));
                    kept += __popcll(mask);
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < FR_CAP / 64; ++c)
                    if (kj[c] >= 0) {
                        w.fr_j[at[c]] = kj[c];
                        w.fr_f[at[c]] = kf[c];
                        w.fr_x[at[c]] = kx[c];
                        w.fr_y[at[c]] = ky[c];
                    }
                if (lane == 0) w.n_fr = kept;
                __syncthreads();
            }

            bool refit = false;
            if (w.n_mem > prm.min_tags_to_curve_fit) {
                // 11 moment sums, one per lane, each over the members in
                // ascending tag order; term = A * B with A in {1, t, t2} and
                // B in {1, t, t2, x, y} (products with 1 are exact)
                const int n = w.n_mem;
                const int selA = lane == 0 || lane == 5 || lane == 8 ? 0
                               : lane == 1 || lane == 6 || lane == 9 ? 1 : 2;
                const int selB = lane <= 2 ? 0 : lane == 3 ? 1 : lane == 4 ? 2
                               : lane <= 7 ? 3 : 4;
                double acc = 0;
                if (lane < 11) {
                    const double fp = p[4];
                    // the coordinate this lane's sum needs, if any
                    const double *coord = selB == 4 ? w.mem_y : w.mem_x;
                    int q = 0;
                    for (; q + 4 <= n; q += 4) {    // loads ahead of the chain
                        double f[4], v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            f[u] = w.mem_f[q + u];
                            v[u] = coord[q + u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double t = f[u] - fp;
                            const double t2 = t * t;
                            const double A = selA == 0 ? 1.0
                                           : selA == 1 ? t : t2;
                            const double B = selB == 0 ? 1.0 : selB == 1 ? t
                                           : selB == 2 ? t2 : v[u];
                            acc += A * B;
                        }
                    }
                    for (; q < n; ++q) {
                        const double t = w.mem_f[q] - fp;
                        const double t2 = t * t;
                        const double A = selA == 0 ? 1.0 : selA == 1 ? t : t2;
                        const double B = selB == 0 ? 1.0 : selB == 1 ? t
                                       : selB == 2 ? t2 : coord[q];
                        acc += A * B;
                    }
                }
                hk::Moments m;
                m.T0 = __shfl(acc, 0, 64);
                m.T1 = __shfl(acc, 1, 64);
                m.T2 = __shfl(acc, 2, 64);
                m.T3 = __shfl(acc, 3, 64);
                m.T4 = __shfl(acc, 4, 64);
                m.X1 = __shfl(acc, 5, 64);
                m.TX = __shfl(acc, 6, 64);
                m.T2X = __shfl(acc, 7, 64);
                m.Y1 = __shfl(acc, 8, 64);
                m.TY = __shfl(acc, 9, 64);
                m.T2Y = __shfl(acc, 10, 64);
                (void)hk::solve_fit5(p, m, prm.rot_s, prm.rot_c);
                refit = true;
            }
            if (p[3] != p[3] || p[3] <= 0) {
                bad = true;
                break;
            }
            if (refit) {
                bool far = false;
                for (int q = lane; q < w.n_mem; q += 64)
                    far = far || sqrt(hk::arc_dist_sq5(p, w.mem_f[q],
                                                       w.mem_x[q], w.mem_y[q],
                                                       prm.rot_s, prm.rot_c)) >
                                     prm.maxd;
                if (__ballot(far) != 0ull) {
                    bad = true;
                    break;
                }
            }
        }
        if (w.give_up) break;
        if (!bad && w.n_mem >= prm.min_tags_per_arc) {
            // eval_arc(requirepeak = false, checkaccel = true)
            bool ok = !(p[0] != p[0] || p[1] != p[1] || p[2] != p[2] ||
                        p[3] != p[3] || p[4] != p[4]);
            if (ok) {
                const double e = p[3];
                if (e < 0) ok = false;
                else if (prm.has_g) {
                    if (2 * e < prm.accel_lo) ok = false;
                    else if (2 * e > prm.accel_hi) ok = false;
                } else if (2 * e < prm.accel_floor) {
                    ok = false;
                }
            }
            if (ok) {
                int close = 0;
                for (int q = lane; q < w.n_mem; q += 64)
                    close += sqrt(hk::arc_dist_sq5(p, w.mem_f[q], w.mem_x[q],
                                                   w.mem_y[q], prm.rot_s,
                                                   prm.rot_c)) < prm.maxd;
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1)
                    close += __shfl_xor(close, s, 64);
                if (close >= prm.min_tags_per_arc) made = 1;
            }
        }
    }

    __syncthreads();
    int status = made ? hk::S3_MADE : hk::S3_NONE;
    if (w.give_up) status = hk::S3_HOST;
    if (status == hk::S3_MADE) {
        unsigned at = 0;
        if (lane == 0) at = atomicAdd(io.made_count, 1u);
        at = __shfl(at, 0, 64);
        if (at >= (unsigned)hk::S3_MADE_CAP) {
            status = hk::S3_HOST;
        } else {
            for (int q = lane; q < w.n_mem; q += 64)
                io.made_members[(long)at * MEM_CAP + q] = w.mem_j[q];
            if (lane == 0) {
                io.made_seed[at] = slot;
                io.made_n[at] = w.n_mem;
                for (int q = 0; q < 5; ++q) io.made_params[at * 5 + q] = p[q];
            }
        }
    }
    for (int q = lane; q < w.n_cons; q += 64)
        io.cons[(long)slot * CONS_CAP + q] = w.cons[q];
    if (lane == 0) {
        io.status[slot] = status;
        io.n_cons[slot] = w.n_cons;
    }
}

// ---------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------
struct Workspace {
    int device = -1;
    hipStream_t stream = nullptr;
    int n_cap = 0, f_cap = 0;
    hipEvent_t event = nullptr;
    int *start = nullptr, *queue = nullptr, *qpos = nullptr, *nb = nullptr,
        *nb_len = nullptr, *flags = nullptr;
    unsigned char *done = nullptr;
    char *out = nullptr;        // one block: status, n_cons, cons, made_*
    char *h_out = nullptr;      // pinned mirror
    unsigned char *h_done = nullptr;    // pinned
    size_t out_bytes = 0;
};

std::mutex g_ws_mutex;
std::vector<Workspace *> g_ws_pool;

void ws_free(Workspace *w) {
    if (!w) return;
    hk::DeviceGuard guard(w->device);
    (void)hipFree(w->start); (void)hipFree(w->queue); (void)hipFree(w->qpos);
    (void)hipFree(w->nb); (void)hipFree(w->nb_len); (void)hipFree(w->flags);
    (void)hipFree(w->done); (void)hipFree(w->out);
    (void)hipHostFree(w->h_out); (void)hipHostFree(w->h_done);
    if (w->event) (void)hipEventDestroy(w->event);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}

// Device / pinned result block: per seed position status, consulted count
// and consulted list; per window the arcs found.
struct OutLayout {
    size_t status, n_cons, cons, made, made_stride, total;
    // inside a window's made block
    size_t m_count, m_seed, m_n, m_params, m_members;
    OutLayout(int n, int windows) {
        size_t at = 0;
        auto take = [&](size_t bytes) {
            const size_t here = at;
            at += (bytes + 63) / 64 * 64;
            return here;
        };
        m_count = take(64);
        m_seed = take(sizeof(int) * hk::S3_MADE_CAP);
        m_n = take(sizeof(int) * hk::S3_MADE_CAP);
        m_params = take(sizeof(double) * 5 * hk::S3_MADE_CAP);
        m_members = take(sizeof(int) * MEM_CAP * hk::S3_MADE_CAP);
        made_stride = at;
        at = 0;
        status = take(sizeof(int) * (size_t)n);
        n_cons = take(sizeof(int) * (size_t)n);
        made = take(made_stride * (size_t)windows);
        cons = take(sizeof(int) * (size_t)CONS_CAP * n);
        total = at;
    }
};

}  // namespace

namespace hk {

struct SeedDevice {
    Workspace *ws = nullptr;
    hk_clip *clip = nullptr;
    SeedParams prm;
    int n = 0, window = 0, windows = 0;
    bool opened_own = false;    // the clip's own EM batch was opened here
};

void seed_release_cached() {
    std::vector<Workspace *> all;
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        all.swap(g_ws_pool);
    }
    for (Workspace *w : all) ws_free(w);
}

void seed_close(SeedDevice *sd) {
    if (!sd) return;
    if (sd->opened_own && sd->clip) em_release(*sd->clip);
    if (sd->ws) {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        if (g_ws_pool.size() < 8) {
            g_ws_pool.push_back(sd->ws);
            sd->ws = nullptr;
        }
    }
    ws_free(sd->ws);
    delete sd;
}

// HK_OK with *out set: ready.  HK_OK with *out null: the device path does not
// apply (no device, lists too long); the caller uses the host code.
int seed_open(hk_clip &c, const SeedParams &prm, const std::vector<int> &queue,
              const std::vector<int> &qpos, int window, SeedDevice **out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || c.device >= ndev) {
        (void)hipGetLastError();
        return HK_OK;
    }
    DeviceGuard guard(c.device);
    if (!guard.ok) return HK_OK;
    Workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        for (size_t i = 0; i < g_ws_pool.size(); ++i)
            if (g_ws_pool[i]->device == c.device) {
                w = g_ws_pool[i];
                g_ws_pool.erase(g_ws_pool.begin() + (long)i);
                break;
            }
    }
    if (!w) {
        w = new Workspace();
        w->device = c.device;
    }
    SeedDevice *sd = new SeedDevice();
    sd->ws = w;
    sd->clip = &c;
    sd->prm = prm;
    sd->n = c.n;
    auto bail = [&](int rc) {
        seed_close(sd);
        return rc;
    };
#define S3_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess)                                                 \
            return bail(hk::fail(HK_ERR_HIP, "%s failed: %s (%s:%d)", #call,  \
                                 hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)
    if (!w->stream)
        S3_HIP(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    const hipStream_t s = w->stream;
    if (c.n > w->n_cap) {
        const int cap = std::max(2 * c.n, 1 << 16);
        (void)hipFree(w->queue); (void)hipFree(w->qpos); (void)hipFree(w->nb);
        (void)hipFree(w->nb_len); (void)hipFree(w->done);
        (void)hipHostFree(w->h_done);
        w->queue = w->qpos = w->nb = w->nb_len = nullptr;
        w->done = w->h_done = nullptr;
        w->n_cap = 0;
        S3_HIP(hipMalloc((void **)&w->queue, sizeof(int) * cap));
        S3_HIP(hipMalloc((void **)&w->qpos, sizeof(int) * cap));
        S3_HIP(hipMalloc((void **)&w->nb, sizeof(int) * (size_t)NB_CAP * cap));
        S3_HIP(hipMalloc((void **)&w->nb_len, sizeof(int) * cap));
        S3_HIP(hipMalloc((void **)&w->done, cap));
        S3_HIP(hipHostMalloc((void **)&w->h_done, cap));
        w->n_cap = cap;
    }
    if (c.F + 1 > w->f_cap) {
        const int cap = std::max(2 * (c.F + 1), 1 << 12);
        (void)hipFree(w->start);
        w->start = nullptr;
        w->f_cap = 0;
        S3_HIP(hipMalloc((void **)&w->start, sizeof(int) * cap));
        w->f_cap = cap;
    }
    if (!w->flags) S3_HIP(hipMalloc((void **)&w->flags, 64));
    sd->window = window;
    sd->windows = (c.n + window - 1) / window;
    {
        const OutLayout lay(c.n, sd->windows);
        if (lay.total > w->out_bytes) {
            (void)hipFree(w->out);
            (void)hipHostFree(w->h_out);
            w->out = w->h_out = nullptr;
            w->out_bytes = 0;
            const size_t cap = lay.total + lay.total / 2;
            S3_HIP(hipMalloc((void **)&w->out, cap));
            S3_HIP(hipHostMalloc((void **)&w->h_out, cap));
            w->out_bytes = cap;
        }
    }
    if (!w->event)
        S3_HIP(hipEventCreateWithFlags(&w->event, hipEventDisableTiming |
                                                      hipEventBlockingSync));
    // tags (in the clip's own EM batch of one, released again when the
    // device side closes: step 4 opens the batch it wants), frame index, seed
    // order
    {
        const bool had_own = c.own != nullptr;
        EmBatch *own = nullptr;
        const int rc = em_own(c, s, &own);
        if (rc != HK_OK) return bail(rc);
        sd->opened_own = !had_own;
    }
    S3_HIP(hipMemcpyAsync(w->start, c.start.data(), sizeof(int) * (c.F + 1),
                          hipMemcpyHostToDevice, s));
    S3_HIP(hipMemcpyAsync(w->queue, queue.data(), sizeof(int) * c.n,
                          hipMemcpyHostToDevice, s));
    S3_HIP(hipMemcpyAsync(w->qpos, qpos.data(), sizeof(int) * c.n,
                          hipMemcpyHostToDevice, s));
    S3_HIP(hipMemsetAsync(w->flags, 0, 64, s));
    const Tags tg{c.own->dev.frame, c.own->dev.x, c.own->dev.y, w->start, c.n,
                  c.F};
    hipLaunchKernelGGL(seed_neighbors_k, dim3((unsigned)c.n), dim3(64), 0, s,
                       tg, prm.gap, prm.has_vmax, prm.v_max, w->nb, w->nb_len,
                       w->flags);
    S3_HIP(hipGetLastError());
    int overflow = 0;
    S3_HIP(hipMemcpyAsync(&overflow, w->flags, sizeof(int),
                          hipMemcpyDeviceToHost, s));
    S3_HIP(hipStreamSynchronize(s));
#undef S3_HIP
    if (overflow) {         // a neighbour list longer than NB_CAP: host code
        seed_close(sd);
        return HK_OK;
    }
    *out = sd;
    return HK_OK;
}

int seed_windows(const SeedDevice *sd) { return sd->windows; }

// Window `index` = seeds queue[index * window ...), against the snapshot
// done[0 .. n).  Blocks (without spinning) until the results are on the host.
int seed_round(SeedDevice *sd, int index, const unsigned char *done,
               SeedRound &r) {
    Workspace *w = sd->ws;
    hk_clip &c = *sd->clip;
    DeviceGuard guard(c.device);
    const hipStream_t s = w->stream;
    if (index < 0 || index >= sd->windows)
        return fail(HK_ERR_INTERNAL, "seed_round: bad window");
    const int p0 = index * sd->window;
    const int count = std::min(sd->window, c.n - p0);
    std::memcpy(w->h_done, done, (size_t)c.n);
    HK_HIP(hipMemcpyAsync(w->done, w->h_done, (size_t)c.n,
                          hipMemcpyHostToDevice, s));
    const OutLayout lay(c.n, sd->windows);
    char *made = w->out + lay.made + lay.made_stride * (size_t)index;
    char *h_made = w->h_out + lay.made + lay.made_stride * (size_t)index;
    HK_HIP(hipMemsetAsync(made + lay.m_count, 0, 64, s));
    SeedIO io;
    io.queue = w->queue;
    io.qpos = w->qpos;
    io.done = w->done;
    io.nb = w->nb;
    io.nb_len = w->nb_len;
    io.p0 = p0;
    io.count = count;
    io.status = (int *)(w->out + lay.status) + p0;
    io.n_cons = (int *)(w->out + lay.n_cons) + p0;
    io.cons = (int *)(w->out + lay.cons) + (size_t)CONS_CAP * p0;
    io.made_count = (unsigned *)(made + lay.m_count);
    io.made_seed = (int *)(made + lay.m_seed);
    io.made_params = (double *)(made + lay.m_params);
    io.made_n = (int *)(made + lay.m_n);
    io.made_members = (int *)(made + lay.m_members);
    const Tags tg{c.own->dev.frame, c.own->dev.x, c.own->dev.y, w->start, c.n,
                  c.F};
    hipLaunchKernelGGL(seed_run_k, dim3((unsigned)count), dim3(64), 0, s, tg,
                       sd->prm, io);
    HK_LAUNCH_CHECK("seed_run_k");
    auto back = [&](size_t off, size_t bytes) {
        return hipMemcpyAsync(w->h_out + off, w->out + off, bytes,
                              hipMemcpyDeviceToHost, s);
    };
    HK_HIP(back(lay.status + sizeof(int) * (size_t)p0, sizeof(int) * count));
    HK_HIP(back(lay.n_cons + sizeof(int) * (size_t)p0, sizeof(int) * count));
    HK_HIP(back(lay.cons + sizeof(int) * (size_t)CONS_CAP * p0,
                sizeof(int) * (size_t)CONS_CAP * count));
    // the arcs found: the fixed part, then as many member lists as there are
    HK_HIP(back(lay.made + lay.made_stride * (size_t)index, lay.m_members));
    HK_HIP(hipEventRecord(w->event, s));
    HK_HIP(hipEventSynchronize(w->event));
    const unsigned found = *(const unsigned *)(h_made + lay.m_count);
    r.made_count = (int)std::min(found, (unsigned)S3_MADE_CAP);
    if (r.made_count > 0) {
        HK_HIP(back(lay.made + lay.made_stride * (size_t)index + lay.m_members,
                    sizeof(int) * (size_t)MEM_CAP * r.made_count));
        HK_HIP(hipEventRecord(w->event, s));
        HK_HIP(hipEventSynchronize(w->event));
    }
    r.p0 = p0;
    r.count = count;
    r.status = (const int *)(w->h_out + lay.status) + p0;
    r.n_cons = (const int *)(w->h_out + lay.n_cons) + p0;
    r.cons = (const int *)(w->h_out + lay.cons) + (size_t)CONS_CAP * p0;
    r.made_seed = (const int *)(h_made + lay.m_seed);
    r.made_params = (const double *)(h_made + lay.m_params);
    r.made_n = (const int *)(h_made + lay.m_n);
    r.made_members = (const int *)(h_made + lay.m_members);
    return HK_OK;
}

}  // namespace hk

#endif  // HK_NO_DEVICE_SEEDS
