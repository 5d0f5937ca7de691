// Steps 3-4 host logic: preliminary arc construction (reference
// tracker.py:450-696, 880-906) and the merge / prune / clean control flow of
// EM_optimize (reference tracker.py:912-990, 1126-1435).  The batched
// arithmetic of step 4 -- E-step, M-step, tilt, member distances -- runs on
// the GPU (hk_em.hip); what stays here is the order-dependent, sequential
// part plus the single-arc trial fits of construct_arcs / merge_arcs.
//
// Tags are indices into SoA arrays ordered by (frame, position in the frame's
// list).  Where the reference iterates Python sets (address order, not
// reproducible) this code iterates in ascending tag index and breaks exact
// ties towards the lower index, like oracle/em_ref.py.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <limits>
#include <mutex>
#include <new>
#include <thread>

#include "hk_clip.h"
#include "hk_seed.h"
#include "hk_simd.h"

namespace hk {

namespace {

constexpr double G_CM = 980.7;

// Python's int(v) for a float that may be far outside int range; results are
// clamped to [0, F) by the callers, so saturating at +-1e9 changes nothing.
long trunc_to_int(double v) {
    if (!(v > -1e9)) return -1000000000L;
    if (!(v < 1e9)) return 1000000000L;
    return (long)v;
}

template <class WeightOf>
Moments moments(const hk_clip &c, const Arc &arc, const std::vector<int> &tags,
                WeightOf weight_of) {
    Moments m;
    for (int i : tags)
        moments_add(m, c.frame[i] - arc.f_peak, weight_of(i), c.x[i], c.y[i]);
    return m;
}

}  // namespace

Rot rotation(const hk_clip &clip) {
    return Rot{std::sin(clip.tilt), std::cos(clip.tilt)};
}

void arc_position(const Arc &arc, double frame, const Rot &rot, double &px,
                  double &py) {
    const double t = frame - arc.f_peak;
    const double xj = arc.a + arc.b * t;
    const double yj = arc.c + arc.e * t * t;
    px = xj * rot.c + yj * rot.s;
    py = yj * rot.c - xj * rot.s;
}

double arc_tag_distance(const hk_clip &clip, const Arc &arc, const Rot &rot,
                        int tag) {
    double px, py;
    arc_position(arc, (double)clip.frame[tag], rot, px, py);
    const double dx = px - clip.x[tag], dy = py - clip.y[tag];
    return std::sqrt(dx * dx + dy * dy);
}

// reference types.py:79-102, quirks included (juggler-coordinate a, b, c
// against screen width/height; int(v + 1) on both ends).
void arc_frame_range(const hk_clip &clip, const Arc &arc, int &f_min,
                     int &f_max) {
    const double F = clip.F;
    double lo_x, hi_x;
    if (arc.b == 0) {
        lo_x = 0;
        hi_x = F;
    } else {
        lo_x = arc.f_peak - arc.a / arc.b;
        hi_x = arc.f_peak + (clip.W - arc.a) / arc.b;
    }
    if (lo_x > hi_x) std::swap(lo_x, hi_x);
    const long ilo_x = trunc_to_int(lo_x + 1), ihi_x = trunc_to_int(hi_x + 1);

    const double q = (clip.H - arc.c) / arc.e;
    double lo_y = q < 0 ? 0.0 : arc.f_peak - std::sqrt(q);
    double hi_y = arc.e < 0 ? F : 2 * arc.f_peak - lo_y;
    if (lo_y > hi_y) std::swap(lo_y, hi_y);
    const long ilo_y = trunc_to_int(lo_y + 1), ihi_y = trunc_to_int(hi_y + 1);

    f_min = (int)std::max(0L, std::max(ilo_x, ilo_y));
    f_max = (int)std::min((long)clip.F - 1, std::min(ihi_x, ihi_y));
}

double median_tag_radius(const hk_clip &clip, const std::vector<int> &tags) {
    std::vector<double> v;
    v.reserve(tags.size());
    for (int i : tags) v.push_back(clip.r[i]);
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    if (n % 2 == 1) return v[n / 2];
    return (v[n / 2 - 1] + v[n / 2]) / 2;
}

bool acceleration_ok(const hk_clip &clip, double e) {
    if (e < 0) return false;
    if (clip.has_g) {
        const double gw = clip.p.g_window;
        if (2 * e < (1 - gw) * clip.g) return false;
        if (2 * e > (1 + gw) * clip.g) return false;
        return true;
    }
    const double max_cm_per_px = clip.p.max_frame_height_cm / clip.H;
    if (2 * e < G_CM / (max_cm_per_px * (clip.fps * clip.fps))) return false;
    return true;
}

namespace {

void solve_fit(Arc &arc, const Moments &m, const Rot &rot);

// Weighted parabola fit of one arc over `tags` (reference tracker.py:713-798).
// Leaves the arc untouched if the normal equations are near singular.
template <class WeightOf>
void fit_one(const hk_clip &c, Arc &arc, const std::vector<int> &tags,
             const Rot &rot, WeightOf weight_of) {
    if (tags.size() < 3) return;
    solve_fit(arc, moments(c, arc, tags, weight_of), rot);
}

// The normal equations of fit_one, given the moment sums.
void solve_fit(Arc &arc, const Moments &m, const Rot &rot) {
    double p[5] = {arc.a, arc.b, arc.c, arc.e, arc.f_peak};
    if (!solve_fit5(p, m, rot.s, rot.c)) return;
    arc.a = p[0];
    arc.b = p[1];
    arc.c = p[2];
    arc.e = p[3];
    arc.f_peak = p[4];
}

// eval_arc with distances computed here (reference tracker.py:800-840).
int eval_arc_host(const hk_clip &c, const Arc &arc, const Rot &rot,
                  bool requirepeak, bool checkaccel) {
    if (std::isnan(arc.a) || std::isnan(arc.b) || std::isnan(arc.c) ||
        std::isnan(arc.e) || std::isnan(arc.f_peak))
        return 1;
    if (arc.tags.empty()) return 5;
    if (requirepeak) {
        int lo = c.frame[arc.tags[0]], hi = lo;
        for (int i : arc.tags) {
            lo = std::min(lo, c.frame[i]);
            hi = std::max(hi, c.frame[i]);
        }
        if (!(lo <= arc.f_peak && arc.f_peak <= hi)) return 3;
    }
    if (checkaccel && !acceleration_ok(c, arc.e)) return 4;
    int close = 0;
    for (int i : arc.tags)
        if (arc_tag_distance(c, arc, rot, i) < c.p.max_distance_pixels) ++close;
    return close < c.p.min_tags_per_arc ? 2 : 0;
}

int report_progress(hk_clip &c) {
    if (c.progress && c.progress(c.progress_user) != 0)
        return fail(HK_ERR_ABORTED, "aborted by the progress callback");
    return HK_OK;
}

// ---------------------------------------------------------------------------
// step 3
// ---------------------------------------------------------------------------

// Worker threads for step 3.  The box's CPU share is what OMP_NUM_THREADS
// says (the GPU boxes export 16); never more than 16.
std::atomic<int> g_host_threads{0};     // hk_set_host_threads(); 0 = auto
std::atomic<int> g_self_check{getenv("HK_SELF_CHECK") != nullptr};

int worker_count() {
    int n = (int)std::thread::hardware_concurrency();
    const int forced = g_host_threads.load();
    if (forced > 0) n = forced;
    else if (const char *e = getenv("HK_THREADS")) n = atoi(e);
    else {
        // one process per GPU: the node's cores are shared by its ranks.
        // torch.distributed.run exports OMP_NUM_THREADS=1 when the variable
        // was unset; with several local ranks 1 is taken to be that default.
        int ranks = 1;
        if (const char *r = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(r));
        n = std::max(1, n / ranks);
        if (const char *o = getenv("OMP_NUM_THREADS")) {
            const int omp = atoi(o);
            if (omp > 1 || (omp == 1 && ranks == 1)) n = std::min(n, omp);
        }
    }
    return std::max(1, std::min(n, 16));
}

inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// construct_arcs (reference tracker.py:487-652).
//
// The reference walks the seed tags (top of the frame first) one at a time.
// A seed changes shared state in exactly one of two ways: it fails, and only
// marks itself done, or -- rarely -- it yields an arc and marks the arc's
// members done.  So seeds are evaluated speculatively, in parallel, each
// under the assumption that every earlier seed that has not been committed
// yet fails.  One thread commits the results in queue order.  A run notes
// every tag whose "not done yet" status it relied on; if an arc accepted
// after the run started took one of those tags, the result is stale and is
// computed again at commit time, when everything before it is final.  The
// outcome is exactly the sequential one.
//
// There is no barrier: workers run up to `window_` seeds ahead of the commit
// position through a ring of result slots.
#ifdef HK_S3_PROF
#include <x86intrin.h>
// Reading the time stamp counter costs tens of cycles (more in a VM), which
// is what the sections measured take themselves: HK_S3_PHASE=k times section k
// alone, so that the run's wall time stays honest and the section's cycles are
// off by its own reads only (a section timed empty gives that overhead).
struct S3Prof {
    unsigned long long cyc[12] = {0}, cnt[12] = {0};
    // how trials end, and the growth steps they took: 0 no tag near the
    // two-point arc, 1 doomed, 2 e <= 0, 3 a member too far after a refit,
    // 4 stopped growing with too few members, 5 eval_arc said no, 6 accepted
    unsigned long long end_n[8] = {0}, end_steps[8] = {0};
    unsigned long long hist[8][8] = {{0}};  // by log2-ish size class
};
static S3Prof g_s3prof;         // the sum over a pass's threads
static const int g_s3_phase = getenv("HK_S3_PHASE") ? atoi(getenv("HK_S3_PHASE")) : -1;
struct S3Scope {
    S3Prof &p;
    const int k;
    const bool on;
    const unsigned long long t0;
    S3Scope(S3Prof &pp, int kk)
        : p(pp), k(kk), on(g_s3_phase < 0 || g_s3_phase == kk),
          t0(on ? __rdtsc() : 0) {}
    ~S3Scope() {
        if (on) p.cyc[k] += __rdtsc() - t0;
        ++p.cnt[k];
    }
};
// (used where `s` is the running thread's Scratch)
#define S3P_SCOPE(k) S3Scope s3p_scope_##k(s.prof, k)
#define S3P_T0(k) const bool s3p_on = g_s3_phase < 0 || g_s3_phase == (k); \
    const unsigned long long s3p_t0 = s3p_on ? __rdtsc() : 0
#define S3P_ADD(k) do { if (s3p_on) s.prof.cyc[k] += __rdtsc() - s3p_t0; ++s.prof.cnt[k]; } while (0)
#define S3P_CNT(k, n) (s.prof.cnt[k] += (n))
#define S3P_END(k, members) do { const unsigned long long st_ = (members) - 2; \
    ++s.prof.end_n[k]; s.prof.end_steps[k] += st_; \
    int cls_ = 0; while (cls_ < 7 && (4ull << cls_) <= st_) ++cls_; \
    s.prof.hist[k][cls_] += st_; } while (0)
#else
#define S3P_SCOPE(k) do {} while (0)
#define S3P_END(k, members) do {} while (0)
#define S3P_T0(k) do {} while (0)
#define S3P_ADD(k) do {} while (0)
#define S3P_CNT(k, n) do {} while (0)
#endif

// Full passes in progress that threads from outside may join
// (hk_step3_help): in a batch most clips are built on one thread each, and
// when the batch runs out the threads that have no clip left help the builds
// still going instead of idling.
class ArcBuilder;
std::mutex g_open_mu;
std::vector<ArcBuilder *> g_open_builders;
constexpr int MAX_SEED_THREADS = 16;    // own workers + helpers per build

class ArcBuilder {
public:
    explicit ArcBuilder(hk_clip &c)
        : c_(c), n_(c.n), queue_((size_t)c.n), qpos_((size_t)c.n),
          nb_ptr_((size_t)c.n, nullptr), nb_len_((size_t)c.n, 0),
          grp_ptr_((size_t)c.n, nullptr), grp_len_((size_t)c.n, 0),
          nb_state_(new std::atomic<uint8_t>[c.n ? c.n : 1]),
          done_(new std::atomic<uint8_t>[c.n ? c.n : 1]),
          taken_(new std::atomic<uint8_t>[c.n ? c.n : 1]),
          done_at_((size_t)c.n, -1) {
        {   // top of the frame first; equal heights keep their tag order
            std::vector<std::pair<double, int>> keyed((size_t)c.n);
            for (int i = 0; i < c.n; ++i) keyed[(size_t)i] = {c.y[i], i};
            bool finite = true;
            for (int i = 0; i < c.n && finite; ++i)
                finite = !std::isnan(c.y[i]);
            if (finite) {
                sort_by_key(keyed);
            } else {            // NaN keys: only a stable sort is defined
                std::stable_sort(keyed.begin(), keyed.end(),
                                 [](const std::pair<double, int> &p,
                                    const std::pair<double, int> &q) {
                                     return p.first < q.first;
                                 });
            }
            for (int k = 0; k < c.n; ++k)
                queue_[(size_t)k] = keyed[(size_t)k].second;
        }
        for (int k = 0; k < n_; ++k) qpos_[(size_t)queue_[(size_t)k]] = k;
        tag_.resize((size_t)c.n);
        for (int i = 0; i < c.n; ++i)
            tag_[(size_t)i] = TagRec{c.x[i], c.y[i], c.frame[i],
                                     qpos_[(size_t)i]};
        // per frame, the tags ordered by x (for the neighbour search)
        xs_idx_.resize((size_t)c.n);
        xs_val_.resize((size_t)c.n);
        std::iota(xs_idx_.begin(), xs_idx_.end(), 0);
        x_index_ok_ = true;
        for (int i = 0; i < c.n && x_index_ok_; ++i)
            x_index_ok_ = std::isfinite(c.x[i]) && std::isfinite(c.y[i]);
        if (x_index_ok_) {
            // (a few dozen tags per frame: insertion sort on the values,
            // equal x keep their tag order)
            for (int k = 0; k < c.n; ++k) xs_val_[(size_t)k] = c.x[k];
            for (int f = 0; f < c.F; ++f) {
                const int a = c.start[f], e = c.start[f + 1];
                if (e - a > 64) {       // crowded frame: the general sort
                    std::sort(xs_idx_.begin() + a, xs_idx_.begin() + e,
                              [&](int p, int q) {
                                  return c.x[p] < c.x[q] ||
                                         (c.x[p] == c.x[q] && p < q);
                              });
                    for (int k = a; k < e; ++k)
                        xs_val_[(size_t)k] = c.x[xs_idx_[(size_t)k]];
                    continue;
                }
                for (int k = a + 1; k < e; ++k) {
                    const double v = xs_val_[(size_t)k];
                    const int id = xs_idx_[(size_t)k];
                    int m = k;
                    while (m > a && xs_val_[(size_t)m - 1] > v) {
                        xs_val_[(size_t)m] = xs_val_[(size_t)m - 1];
                        xs_idx_[(size_t)m] = xs_idx_[(size_t)m - 1];
                        --m;
                    }
                    xs_val_[(size_t)m] = v;
                    xs_idx_[(size_t)m] = id;
                }
            }
        }
        nthreads_ = std::min(worker_count(), MAX_SEED_THREADS);
        scratch_.reset(new Scratch[(size_t)MAX_SEED_THREADS]);
        for (int t = 0; t < nthreads_; ++t) scratch_[t].init(c.n, c.F);
        for (int t = 1; t < nthreads_; ++t)
            pool_.emplace_back([this, t] { worker_loop(scratch_[t]); });
    }

    ~ArcBuilder() {
        quit_.store(true);
        generation_.fetch_add(1);
        for (std::thread &t : pool_) t.join();
    }

    // keyed (no NaN keys, second = 0, 1, 2 ...) into the order std::sort
    // gives pairs: by key, equal keys (-0.0 == 0.0 among them) by index.  A
    // stable byte-wise radix sort on the keys' bit patterns, made monotonic;
    // passes in which all keys share the byte are skipped (pixel coordinates
    // differ in few of them): 0.3 ms for 50 000 tags against 2.5 ms.
    static void sort_by_key(std::vector<std::pair<double, int>> &keyed) {
        const size_t n = keyed.size();
        if (n < 256) {
            std::sort(keyed.begin(), keyed.end());
            return;
        }
        std::vector<uint64_t> key(n), key2(n);
        std::vector<int> idx(n), idx2(n);
        for (size_t i = 0; i < n; ++i) {
            const double v = keyed[i].first + 0.0;      // -0.0 -> +0.0
            uint64_t u;
            std::memcpy(&u, &v, sizeof(u));
            key[i] = (u >> 63) ? ~u : (u | (1ull << 63));
            idx[i] = keyed[i].second;
        }
        for (int shift = 0; shift < 64; shift += 8) {
            size_t count[257] = {0};
            for (size_t i = 0; i < n; ++i) ++count[((key[i] >> shift) & 255) + 1];
            bool one = false;
            for (int b = 1; b <= 256; ++b) one = one || count[b] == n;
            if (one) continue;
            for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
            for (size_t i = 0; i < n; ++i) {
                const size_t at = count[(key[i] >> shift) & 255]++;
                key2[at] = key[i];
                idx2[at] = idx[i];
            }
            key.swap(key2);
            idx.swap(idx2);
        }
        for (size_t i = 0; i < n; ++i) {
            // only the order is used afterwards; keep the pair's key right
            keyed[i].second = idx[i];
        }
    }

    // One pass over all seeds with the clip's current gravity estimate;
    // stops after `maxcount` arcs if maxcount > 0.
    int construct(int maxcount, std::vector<Arc> &out) {
        hk_clip &c = c_;
        has_vmax_ = c.has_g;
        v_max_ = has_vmax_ ? std::sqrt(2 * c.g * c.H) : 0.0;
        maxd_ = c.p.max_distance_pixels;
        maxd_sq_lo_ = maxd_ * maxd_ * (1 - 1e-12);
        maxd_sq_hi_ = maxd_ * maxd_ * (1 + 1e-12);
        rot_ = rotation(c);
        // what eval_arc() will ask of a finished trial arc's e
        // (acceleration_ok), for the early exit of doomed trials
        if (c.has_g) {
            e_lo_ = 0.5 * (1 - c.p.g_window) * c.g;
            e_hi_ = 0.5 * (1 + c.p.g_window) * c.g;
        } else {
            const double max_cm_per_px = c.p.max_frame_height_cm / c.H;
            e_lo_ = 0.5 * G_CM / (max_cm_per_px * (c.fps * c.fps));
            e_hi_ = std::numeric_limits<double>::infinity();
        }
        prune_ok_ = rot_.s == 0.0 && rot_.c == 1.0 && std::isfinite(e_lo_) &&
                    !getenv("HK_STEP3_NO_PRUNE");
        if (c.has_g) {
            e0_ = 0.5 * c.g;
        } else {
            const double cm_per_px = c.p.default_frame_height_cm / c.H;
            e0_ = 0.5 * G_CM / (cm_per_px * (c.fps * c.fps));
        }
        for (int i = 0; i < n_; ++i) {
            nb_state_[i].store(0, std::memory_order_relaxed);
            done_[i].store(0, std::memory_order_relaxed);
            taken_[i].store(0, std::memory_order_relaxed);
        }
        for (int t = 0; t < MAX_SEED_THREADS; ++t) scratch_[t].pool_reset();

        const bool use_device = maxcount <= 0 && has_vmax_ && x_index_ok_ &&
                                device_wanted();
        // the trial pass stops after a handful of arcs: keep its speculation
        // shallow so that little work is thrown away.  With device results
        // most seeds cost the workers nothing, so a short window keeps them
        // busy and few results go stale before they are committed.
        int per_thread = use_device ? 32 : 256;
        if (const char *e = getenv("HK_STEP3_WINDOW")) per_thread = atoi(e);
        size_t want = maxcount > 0 ? (size_t)nthreads_
                                   : (size_t)std::max(8, per_thread * nthreads_);
        if (nthreads_ == 1) want = 1;
        // helpers may join a full pass: leave them room to run ahead
        const bool joinable = maxcount <= 0 && n_ >= 1024;
        if (joinable) want = std::max(want, (size_t)(per_thread * MAX_SEED_THREADS));
        window_ = 1;
        while (window_ < want) window_ <<= 1;
        slots_.reset(new Slot[window_]);
        std::fill(done_at_.begin(), done_at_.end(), -1);
        n_accepted_ = 0;
        accepted_.store(0);
        accepting_.store(0);
        next_.store(0);
        // (also for a one-thread build: same time on its own, and the
        // threads that come to help find the lists made)
        build_next_.store(maxcount <= 0 && has_vmax_ &&
                                  !getenv("HK_STEP3_LAZY_LISTS")
                              ? 0 : n_);
        commit_pos_.store(0);
        stop_.store(false);
        stat_stale_ = stat_helped_ = 0;
        stat_stale_ms_ = 0;

        // In the full pass the GPU evaluates windows of seeds ahead of the
        // workers (hk_seed.hip); a worker that claims a seed whose device
        // result is in takes that instead of running the seed.  The trial
        // pass is short and stays on the host.
        dev_ = nullptr;
        dev_error_.store(HK_OK);
        std::thread feeder;
        if (use_device) {
            const int rc = device_start();
            if (rc != HK_OK) return rc;
            if (dev_) feeder = std::thread([this] { feeder_loop(); });
            // one window for the whole clip: nothing to do here before it
            // is in (in a batch the other clips keep the cores busy)
            if (dev_ && dev_window_ >= n_)
                while (!dev_rounds_[0].ready.load(std::memory_order_acquire) &&
                       dev_error_.load() == HK_OK)
                    std::this_thread::yield();
        }

        active_.store(nthreads_ - 1);
        generation_.fetch_add(1);               // release the workers
        if (joinable) {
            std::lock_guard<std::mutex> lock(g_open_mu);
            helper_slots_used_ = 0;
            g_open_builders.push_back(this);
        }

        Scratch &mine = scratch_[0];
        const size_t Q = (size_t)n_;
        int rc = HK_OK;
        int since_progress = 0;
        for (size_t k = 0; k < Q; ++k) {
            const int t1 = queue_[k];
            Slot &slot = slots_[k & (window_ - 1)];
            {   // the slots were written by other cores: have the lines of
                // the next ones on their way while this one is dealt with
                __builtin_prefetch(&slots_[(k + 12) & (window_ - 1)]);
                const Slot &soon = slots_[(k + 6) & (window_ - 1)];
                if (soon.state.load(std::memory_order_acquire) == 1)
                    __builtin_prefetch(soon.res.consulted.data());
            }
            if (done_[t1].load(std::memory_order_relaxed)) {
                // joined an arc meanwhile; whoever claimed k saw that too,
                // or will; wait for the slot so that it can be recycled
                wait_slot(slot, k, mine);
            } else {
                wait_slot(slot, k, mine);
                SeedResult &r = slot.res;
                bool stale = slot.skipped;
                if (!stale && slot.epoch < n_accepted_)
                    for (int j : r.consulted)
                        if (done_at_[(size_t)j] >= (long)slot.epoch) {
                            stale = true;
                            break;
                        }
                if (stale) {
                    ++stat_stale_;
                    const auto ts = std::chrono::steady_clock::now();
                    run_seed(t1, (int)k, mine, r);
                    stat_stale_ms_ += std::chrono::duration<double, std::milli>(
                        std::chrono::steady_clock::now() - ts).count();
                }
                if (r.made) {
                    accepting_.store(n_accepted_ + 1, std::memory_order_release);
                    for (int j : r.arc.tags) {
                        done_[j].store(1, std::memory_order_relaxed);
                        taken_[j].store(1, std::memory_order_relaxed);
                        done_at_[(size_t)j] = (long)n_accepted_;
                    }
                    ++n_accepted_;
                    accepted_.store(n_accepted_, std::memory_order_release);
                    out.push_back(std::move(r.arc));
                    r.made = false;
                }
                // (a seed that made no arc is "done" by its place in the
                // queue -- is_done() -- and needs no flag: a store per seed
                // into the flags every worker reads all the time cost a
                // 16-thread build a third of its seed time in cache misses)
                if (++since_progress >= 256) {
                    since_progress = 0;
                    rc = report_progress(c);
                }
            }
            slot.state.store(0, std::memory_order_relaxed);
            commit_pos_.store(k + 1, std::memory_order_release);
            if (rc != HK_OK ||
                (maxcount > 0 && (int)out.size() >= maxcount))
                break;
        }
        if (joinable) {     // no new helpers from here on
            std::lock_guard<std::mutex> lock(g_open_mu);
            g_open_builders.erase(std::find(g_open_builders.begin(),
                                            g_open_builders.end(), this));
        }
        stop_.store(true, std::memory_order_release);
        {
            int spins = 0;
            while (active_.load(std::memory_order_acquire) > 0) {
                if (++spins > 2000) {
                    std::this_thread::yield();
                    spins = 0;
                } else {
                    cpu_relax();
                }
            }
        }
        if (feeder.joinable()) feeder.join();
        if (dev_) {
            seed_close(dev_);
            dev_ = nullptr;
            if (rc == HK_OK && dev_error_.load() != HK_OK)
                rc = fail(dev_error_.load(), "step 3 on the device: %s",
                          dev_message_);
        }
        stat_busy_ms_ = 0;
        stat_steps_ = stat_runs_ = 0;
        for (int t = 0; t < MAX_SEED_THREADS; ++t) {
            stat_busy_ms_ += scratch_[t].busy_ms;
            stat_steps_ += scratch_[t].steps;
            stat_runs_ += scratch_[t].runs;
            scratch_[t].busy_ms = 0;
            scratch_[t].steps = scratch_[t].runs = 0;
#ifdef HK_S3_PROF
            for (int k = 0; k < 12; ++k) {
                g_s3prof.cyc[k] += scratch_[t].prof.cyc[k];
                g_s3prof.cnt[k] += scratch_[t].prof.cnt[k];
            }
            for (int k = 0; k < 8; ++k) {
                g_s3prof.end_n[k] += scratch_[t].prof.end_n[k];
                g_s3prof.end_steps[k] += scratch_[t].prof.end_steps[k];
                for (int h = 0; h < 8; ++h)
                    g_s3prof.hist[k][h] += scratch_[t].prof.hist[k][h];
            }
            scratch_[t].prof = S3Prof();
#endif
        }
        return rc;
    }

    // Work left per thread on it: what a helper looks at to choose a build.
    // Call with g_open_mu held.
    double load_per_thread() const {
        const size_t done = commit_pos_.load(std::memory_order_relaxed);
        const int threads = nthreads_ + __builtin_popcount(helper_slots_used_);
        return (double)((size_t)n_ - std::min(done, (size_t)n_)) / threads;
    }

    // Takes a free scratch slot for a helper; call with g_open_mu held (the
    // builder is in g_open_builders, so its pass is running and it will wait
    // for active_ to drop before it ends).  -1 if all slots are taken.
    int helper_join() {
        for (int t = nthreads_; t < MAX_SEED_THREADS; ++t)
            if (!(helper_slots_used_ & (1u << t))) {
                helper_slots_used_ |= 1u << t;
                active_.fetch_add(1, std::memory_order_acq_rel);
                return t;
            }
        return -1;
    }

    // The calling thread runs seeds of this pass until there is none left to
    // claim or the pass ends.  After the final fetch_sub the builder may be
    // gone: nothing of it is touched afterwards.
    void helper_work(int slot, int max_seeds) {
        Scratch &s = scratch_[slot];
        if ((int)s.marks.size() != n_) s.init(c_.n, c_.F);
        int idle = 0, ran = 0;
        while (!stop_.load(std::memory_order_acquire) &&
               next_.load(std::memory_order_relaxed) < (size_t)n_ &&
               ran < max_seeds) {
            if (run_next(s)) {
                idle = 0;
                ++ran;
            } else if (++idle > 2000) {
                std::this_thread::yield();
                idle = 0;
            } else {
                cpu_relax();
            }
        }
        {
            std::lock_guard<std::mutex> lock(g_open_mu);
            helper_slots_used_ &= ~(1u << slot);
        }
        active_.fetch_sub(1, std::memory_order_release);
    }

    // Seeds on the GPU are opt-in (HK_STEP3_DEVICE=1; 2 = whatever the clip's
    // size, for tests).  Measured on the 1080p60 benchmark clip (MI355X + 16
    // host cores): the device results are exact and spare the host 41 % of
    // its seed time, but one wave needs ~10 us per growth step against
    // ~0.24 us on a host core, so a clip alone gets slower (the long seeds
    // gate the launch: 11 ms) and a pipelined batch gains only ~8 %.
    bool device_wanted() const {
        const char *e = getenv("HK_STEP3_DEVICE");
        if (!e) return false;
        if (atoi(e) == 2) return true;
        return atoi(e) == 1 && n_ >= 4096;
    }

    // (statistics the committer updates as it goes: a cache line of their
    // own, away from the members every worker reads all the time)
    alignas(128) std::atomic<long> stat_dev_used_{0}, stat_dev_host_{0};

    long stat_stale_ = 0, stat_helped_ = 0, stat_steps_ = 0, stat_runs_ = 0;
    double stat_stale_ms_ = 0, stat_busy_ms_ = 0;
    bool collect_stats_ = false;       // HK_STEP3_STATS: time every seed

private:
    // neighbour lists live in per-thread chunked pools (no malloc per tag)
    struct alignas(128) Scratch {   // own cache lines: no false sharing
        // per tag: run stamp of `consulted`, trial stamp of the frontier
        struct Marks {
            int seen, in_frontier;
        };
        std::vector<Marks> marks;
        std::vector<int> taken, cand, order;
        std::vector<double> vel;
        // the tags the current run found "not done", collected here and
        // handed to the result slot in one piece when the run ends: the slot
        // is memory the committer polls and another thread wrote last
        std::vector<int> consulted;
        // The members of the trial arc, ascending tag order, with their
        // frame, x, y alongside (what the fits and checks run over): they
        // occupy [m_lo, m_hi) of arrays with room on both sides, because an
        // arc mostly grows at its ends -- a new first member moves m_lo, not
        // the members.
        std::vector<int> mem_id;
        std::vector<double> mem_f, mem_x, mem_y;
        int m_lo = 0, m_hi = 0, m_centre = 0;
        // The frontier: tag, frame, x, y (no gathers in the distance loop)
        // and the squared distance of the last scan, nf entries in use.
        std::vector<int> fr_id;
        std::vector<double> fr_f, fr_x, fr_y, fr_q;
        size_t nf = 0;
        int stamp = 0, run_stamp = 0;
        double busy_ms = 0;
        long steps = 0, runs = 0;       // growth steps / seeds run (statistics)
#ifdef HK_S3_PROF
        S3Prof prof;
#endif
        std::vector<std::unique_ptr<int[]>> chunks;
        size_t chunk_used = 0, chunk_index = 0;
        static constexpr size_t CHUNK = 1 << 16;
        void init(int n, int F) {
            marks.assign((size_t)n, Marks{0, 0});
            taken.assign((size_t)(F > 0 ? F : 1), 0);
            // at most one member per frame; every insertion moves one end
            // of the range by one, so F + 2 on either side is never left
            m_centre = (F > 0 ? F : 1) + 2;
            const size_t cap = 2 * (size_t)m_centre + 8;
            mem_id.assign(cap, 0);
            mem_f.assign(cap, 0.0);
            mem_x.assign(cap, 0.0);
            mem_y.assign(cap, 0.0);
            if (fr_id.empty()) fr_resize(256);
        }
        void fr_resize(size_t cap) {
            fr_id.resize(cap);
            fr_f.resize(cap);
            fr_x.resize(cap);
            fr_y.resize(cap);
            fr_q.resize(cap);
        }
        void fr_push(int j, double f, double x, double y) {
            if (nf == fr_id.size()) fr_resize(2 * nf);
            fr_id[nf] = j;
            fr_f[nf] = f;
            fr_x[nf] = x;
            fr_y[nf] = y;
            ++nf;
        }
        int member_count() const { return m_hi - m_lo; }
        void members_clear() { m_lo = m_hi = m_centre; }
        // keeps the ascending tag order (j is not a member yet)
        void member_insert(int j, double f, double x, double y) {
            int *id = mem_id.data();
            int at;
            if (m_hi == m_lo || j > id[m_hi - 1]) {
                at = m_hi++;
            } else if (j < id[m_lo]) {
                at = --m_lo;
            } else {
                const int pos = (int)(std::upper_bound(id + m_lo, id + m_hi, j) - id);
                if (pos - m_lo < m_hi - pos) {      // the front part is shorter
                    const size_t cnt = (size_t)(pos - m_lo);
                    std::memmove(id + m_lo - 1, id + m_lo, cnt * sizeof(int));
                    for (std::vector<double> *v : {&mem_f, &mem_x, &mem_y})
                        std::memmove(v->data() + m_lo - 1, v->data() + m_lo,
                                     cnt * sizeof(double));
                    --m_lo;
                    at = pos - 1;
                } else {
                    const size_t cnt = (size_t)(m_hi - pos);
                    std::memmove(id + pos + 1, id + pos, cnt * sizeof(int));
                    for (std::vector<double> *v : {&mem_f, &mem_x, &mem_y})
                        std::memmove(v->data() + pos + 1, v->data() + pos,
                                     cnt * sizeof(double));
                    ++m_hi;
                    at = pos;
                }
            }
            id[at] = j;
            mem_f[(size_t)at] = f;
            mem_x[(size_t)at] = x;
            mem_y[(size_t)at] = y;
        }
        void pool_reset() {
            chunk_index = 0;
            chunk_used = 0;
        }
        int *pool_alloc(size_t len) {
            if (len > CHUNK) {      // never for real inputs; keep it correct
                chunks.insert(chunks.begin() + (long)chunk_index,
                              std::unique_ptr<int[]>(new int[len]));
                return chunks[chunk_index++].get();
            }
            if (chunks.size() <= chunk_index || chunk_used + len > CHUNK) {
                if (chunks.size() > chunk_index && chunk_used > 0)
                    ++chunk_index;
                if (chunks.size() <= chunk_index)
                    chunks.emplace_back(new int[CHUNK]);
                chunk_used = 0;
            }
            int *p = chunks[chunk_index].get() + chunk_used;
            chunk_used += len;
            return p;
        }
    };
    // Memory in whole cache lines of its own: the results of neighbouring
    // queue positions are written by different threads at the same time.
    template <class T>
    struct LineAlloc {
        using value_type = T;
        LineAlloc() = default;
        template <class U>
        LineAlloc(const LineAlloc<U> &) {}
        T *allocate(size_t n) {
            const size_t bytes = (n * sizeof(T) + 127) / 128 * 128;
            return static_cast<T *>(::operator new(bytes, std::align_val_t(128)));
        }
        void deallocate(T *p, size_t) {
            ::operator delete(p, std::align_val_t(128));
        }
        template <class U>
        bool operator==(const LineAlloc<U> &) const { return true; }
        template <class U>
        bool operator!=(const LineAlloc<U> &) const { return false; }
    };
    struct SeedResult {
        bool made = false;
        // tags the run found "not done" and may have used for that reason
        std::vector<int, LineAlloc<int>> consulted;
        Arc arc;
    };
    struct alignas(128) Slot {
        std::atomic<int> state{0};      // 0 empty, 1 result ready
        bool skipped = false;           // the seed was already done
        size_t epoch = 0;               // arcs accepted when the run started
        SeedResult res;
    };
    struct Span {
        const int *p;
        int len;
        const int *begin() const { return p; }
        const int *end() const { return p + len; }
    };
    struct DevRound {               // one window of seeds run on the device
        std::atomic<int> ready{0};
        size_t epoch = 0;           // arcs accepted when its flags were taken
        SeedRound r;
        std::vector<int> made_of;   // position in the window -> made_* index
    };

    static bool same_result(const SeedResult &a, const SeedResult &b) {
        if (a.made != b.made) return false;
        if (!a.made) return true;
        const double pa[5] = {a.arc.a, a.arc.b, a.arc.c, a.arc.e, a.arc.f_peak};
        const double pb[5] = {b.arc.a, b.arc.b, b.arc.c, b.arc.e, b.arc.f_peak};
        return a.arc.tags == b.arc.tags &&
               std::memcmp(pa, pb, sizeof(pa)) == 0;
    }

    int device_start() {
        hk_clip &c = c_;
        SeedParams prm;
        prm.e0 = e0_;
        prm.maxd = maxd_;
        prm.rot_s = rot_.s;
        prm.rot_c = rot_.c;
        prm.v_max = v_max_;
        prm.has_g = c.has_g ? 1 : 0;
        prm.has_vmax = has_vmax_ ? 1 : 0;
        prm.accel_lo = (1 - c.p.g_window) * c.g;
        prm.accel_hi = (1 + c.p.g_window) * c.g;
        {
            const double max_cm_per_px = c.p.max_frame_height_cm / c.H;
            prm.accel_floor = G_CM / (max_cm_per_px * (c.fps * c.fps));
        }
        prm.gap = c.p.max_frame_gap_in_arc;
        prm.min_tags_to_curve_fit = c.p.min_tags_to_curve_fit;
        prm.min_tags_per_arc = c.p.min_tags_per_arc;
        // By default the whole clip is one window, run against the flags of
        // the start, every seed to its end: most of the long-running seeds
        // follow clutter tracks that no arc ever touches, so their results
        // still hold when the committer gets to them.
        prm.budget = 1 << 20;
        if (const char *e = getenv("HK_STEP3_DEVICE_BUDGET"))
            prm.budget = std::max(1, atoi(e));
        dev_window_ = std::max(n_, 64);
        if (const char *e = getenv("HK_STEP3_DEVICE_WINDOW"))
            dev_window_ = std::max(64, atoi(e));
        const int rc = seed_open(c, prm, queue_, qpos_, dev_window_, &dev_);
        if (rc != HK_OK || !dev_) return rc;    // no device: host only
        const int nw = seed_windows(dev_);
        dev_rounds_.reset(new DevRound[(size_t)nw]);
        stat_dev_used_.store(0);
        stat_dev_host_.store(0);
        return HK_OK;
    }

    // Launches window after window, each against the taken flags of the
    // moment, never more than two windows ahead of the committer (a result
    // computed long before its seed is committed is more likely stale) and
    // never for seeds the workers have passed already.
    void feeder_loop() {
        const int nw = seed_windows(dev_);
        std::vector<unsigned char> snap((size_t)n_);
        for (int r = 0; r < nw; ++r) {
            const size_t p0 = (size_t)r * (size_t)dev_window_;
            int idle = 0;
            while (!stop_.load(std::memory_order_acquire) &&
                   p0 > commit_pos_.load(std::memory_order_acquire) +
                            2 * (size_t)dev_window_) {
                if (++idle > 200) {
                    std::this_thread::yield();
                    idle = 0;
                } else {
                    cpu_relax();
                }
            }
            if (stop_.load(std::memory_order_acquire)) return;
            if (p0 + (size_t)dev_window_ <=
                next_.load(std::memory_order_relaxed))
                continue;                       // the workers were faster
            DevRound &d = dev_rounds_[(size_t)r];
            // epoch first: a tag taken while the snapshot is copied may or
            // may not be in it, and then counts as taken after the epoch
            d.epoch = accepted_.load(std::memory_order_acquire);
            for (int i = 0; i < n_; ++i)
                snap[(size_t)i] = done_[i].load(std::memory_order_relaxed);
            const int rc = seed_round(dev_, r, snap.data(), d.r);
            if (rc != HK_OK) {
                snprintf(dev_message_, sizeof(dev_message_), "%s",
                         error_buffer());
                dev_error_.store(rc);
                return;
            }
            d.made_of.assign((size_t)d.r.count, -1);
            for (int m = 0; m < d.r.made_count; ++m)
                d.made_of[(size_t)d.r.made_seed[m]] = m;
            d.ready.store(1, std::memory_order_release);
        }
    }

    // The device's result for queue position p, if there is one that counts.
    bool take_device_result(size_t p, Slot &slot) {
        DevRound &d = dev_rounds_[p / (size_t)dev_window_];
        if (!d.ready.load(std::memory_order_acquire)) return false;
        const int k = (int)(p - (size_t)d.r.p0);
        const int st = d.r.status[k];
        if (st != S3_NONE && st != S3_MADE) {
            stat_dev_host_.fetch_add(1, std::memory_order_relaxed);
            return false;
        }
        // The device saw the flags of an earlier moment.  Its result holds
        // now if none of the tags it looked at has been taken since (then it
        // is the result a run started now would give); later acceptances are
        // the committer's staleness test, from this epoch on.
        SeedResult &res = slot.res;
        const int *cons = d.r.cons + (size_t)k * S3_CONSULTED;
        const size_t epoch = accepted_.load(std::memory_order_acquire);
        for (int q = 0; q < d.r.n_cons[k]; ++q)
            if (done_at_now(cons[q])) {
                stat_dev_host_.fetch_add(1, std::memory_order_relaxed);
                return false;
            }
        res.consulted.assign(cons, cons + d.r.n_cons[k]);
        res.made = st == S3_MADE;
        if (res.made) {
            const int m = d.made_of[(size_t)k];
            const double *pp = d.r.made_params + 5 * (size_t)m;
            res.arc = Arc();
            res.arc.a = pp[0];
            res.arc.b = pp[1];
            res.arc.c = pp[2];
            res.arc.e = pp[3];
            res.arc.f_peak = pp[4];
            const int *mem = d.r.made_members + (size_t)m * S3_MEMBERS;
            res.arc.tags.assign(mem, mem + d.r.made_n[m]);
        }
        slot.epoch = epoch;
        stat_dev_used_.fetch_add(1, std::memory_order_relaxed);
        return true;
    }


    // tag j belongs to an accepted arc (committed seeds that made no arc are
    // "done" only in the sense that they come earlier in the queue)
    bool done_at_now(int j) const {
        return taken_[j].load(std::memory_order_acquire) != 0;
    }

    bool is_done(int j, int pos) const {
        return qpos_[(size_t)j] < pos ||
               done_[j].load(std::memory_order_relaxed);
    }

    // Claims the next queue position if it lies inside the window and runs
    // it.  Returns false if there was nothing to claim.
    bool run_next(Scratch &s) {
        if (build_next_.load(std::memory_order_relaxed) < n_ && build_some(s))
            return true;
        size_t p = next_.load(std::memory_order_relaxed);
        for (;;) {
            if (p >= (size_t)n_ ||
                p >= commit_pos_.load(std::memory_order_acquire) + window_)
                return false;
            if (next_.compare_exchange_weak(p, p + 1,
                                            std::memory_order_relaxed))
                break;
        }
        Slot &slot = slots_[p & (window_ - 1)];
        const int t1 = queue_[p];
        if (done_[t1].load(std::memory_order_relaxed)) {
            slot.skipped = true;
        } else if (dev_ && take_device_result(p, slot)) {
            slot.skipped = false;
            if (g_self_check.load(std::memory_order_relaxed)) {
                // the host's own result against the same flags: what the
                // device saw as taken is a subset of what is taken now, so
                // only compare when nothing was accepted since
                SeedResult mine;
                const size_t before = slot.epoch;
                run_seed(t1, (int)p, s, mine);
                if (accepting_.load(std::memory_order_acquire) == before &&
                    !same_result(mine, slot.res)) {
                    snprintf(dev_message_, sizeof(dev_message_),
                             "self check: device and host disagree on seed "
                             "%zu (tag %d): host %s, device %s", p, t1,
                             mine.made ? "arc" : "no arc",
                             slot.res.made ? "arc" : "no arc");
                    dev_error_.store(HK_ERR_INTERNAL);
                }
            }
        } else {
            slot.skipped = false;
            slot.epoch = accepted_.load(std::memory_order_acquire);
            if (!collect_stats_) {
                run_seed(t1, (int)p, s, slot.res);
            } else {
                const auto tb = std::chrono::steady_clock::now();
                run_seed(t1, (int)p, s, slot.res);
                s.busy_ms += std::chrono::duration<double, std::milli>(
                    std::chrono::steady_clock::now() - tb).count();
            }
        }
        slot.state.store(1, std::memory_order_release);
        return true;
    }

    // The neighbour lists of a run of tags (full pass).  Every list is
    // needed sooner or later; built on first use, each list is written by
    // whichever thread got there first, and the per-tag tables those writes
    // go to are read by all threads all the time -- with 16 threads on one
    // clip the seeds took 1.8 x the time they take on one thread.  Built in
    // runs of consecutive tags by the threads of the pass before they turn to
    // the seeds, each thread writes its own stretch of the tables, once.
    // (A seed that needs a list nobody has built yet still builds it.)
    bool build_some(Scratch &s) {
        constexpr int RUN = 128;
        const int i0 = build_next_.fetch_add(RUN, std::memory_order_relaxed);
        if (i0 >= n_) return false;
        const int i1 = std::min(n_, i0 + RUN);
        for (int i = i0; i < i1; ++i) {
            uint8_t expected = 0;
            if (nb_state_[i].compare_exchange_strong(expected, 1)) {
                build_neighbors(i, s);
                nb_state_[i].store(2, std::memory_order_release);
            }
        }
        return true;
    }

    // Committer: until slot k is ready, lend a hand.
    void wait_slot(Slot &slot, size_t k, Scratch &mine) {
        (void)k;
        while (slot.state.load(std::memory_order_acquire) != 1) {
            if (run_next(mine)) ++stat_helped_;
            else cpu_relax();
        }
    }

    void worker_loop(Scratch &s) {
        unsigned seen = 0;
        for (;;) {
            unsigned g;
            int spins = 0;
            while ((g = generation_.load(std::memory_order_acquire)) == seen) {
                if (++spins > 4000) {
                    std::this_thread::yield();
                    spins = 0;
                } else {
                    cpu_relax();
                }
            }
            seen = g;
            if (quit_.load()) return;
            int idle = 0;
            while (!stop_.load(std::memory_order_acquire)) {
                if (run_next(s)) {
                    idle = 0;
                } else if (++idle > 2000) {
                    std::this_thread::yield();
                    idle = 0;
                } else {
                    cpu_relax();
                }
            }
            active_.fetch_sub(1, std::memory_order_release);
        }
    }

    // Neighbours of tag i: every tag within max_frame_gap_in_arc + 1 frames,
    // other frames only, ordered by the speed needed to get there (stable),
    // capped at v_max once g is known (reference tracker.py:654-696).  Built
    // on first use by whichever thread needs it; the reference builds all of
    // them up front, which yields the same lists.
    Span neighbors(int i, Scratch &s) {
        uint8_t st = nb_state_[i].load(std::memory_order_acquire);
        if (st != 2) {
            uint8_t expected = 0;
            if (nb_state_[i].compare_exchange_strong(expected, 1)) {
                build_neighbors(i, s);
                nb_state_[i].store(2, std::memory_order_release);
            } else {
                while (nb_state_[i].load(std::memory_order_acquire) != 2)
                    cpu_relax();
            }
        }
        return Span{nb_ptr_[(size_t)i], nb_len_[(size_t)i]};
    }

    void build_neighbors(int i, Scratch &s) {
        const hk_clip &c = c_;
        const int f = c.frame[i];
        const int gap = c.p.max_frame_gap_in_arc;
        const int lo = std::max(0, f - gap - 1);
        const int hi = std::min(c.F, f + gap + 2);
        s.cand.clear();
        s.vel.clear();
        if (has_vmax_ && x_index_ok_) {
            // with a speed cap only tags within v_max * |df| can qualify:
            // visit just the x-window of every frame (tags sorted by x per
            // frame), then order by (speed, tag index), which is what the
            // stable sort below gives for candidates met in index order
            const double xi = c.x[i], yi = c.y[i];
            for (int g = lo; g < hi; ++g) {
                if (g == f) continue;
                const int df = std::abs(g - f);
                const double r = v_max_ * df * (1 + 1e-12) + 1e-9;
                // farther than this (squared) is beyond the cap whatever the
                // root and the division round to: most tags of the x-window
                // are, and need neither
                const double far_sq = (v_max_ * df) * (v_max_ * df) * (1 + 1e-9);
                const double *xb = xs_val_.data() + c.start[g];
                const double *xe = xs_val_.data() + c.start[g + 1];
                for (const double *p = std::lower_bound(xb, xe, xi - r);
                     p < xe && *p <= xi + r; ++p) {
                    const int j = xs_idx_[(size_t)(p - xs_val_.data())];
                    const double dx = c.x[j] - xi, dy = c.y[j] - yi;
                    const double q = dx * dx + dy * dy;
                    if (q > far_sq) continue;
                    const double v = std::sqrt(q) / df;
                    if (!(v <= v_max_)) continue;
                    s.cand.push_back(j);
                    s.vel.push_back(v);
                }
            }
            const size_t m = s.cand.size();
            s.order.resize(m);
            std::iota(s.order.begin(), s.order.end(), 0);
            std::sort(s.order.begin(), s.order.end(), [&](int p, int q) {
                return s.vel[p] < s.vel[q] ||
                       (s.vel[p] == s.vel[q] && s.cand[p] < s.cand[q]);
            });
        } else {
            for (int j = c.start[lo]; j < c.start[hi]; ++j) {
                if (c.frame[j] == f) continue;
                const double dx = c.x[j] - c.x[i], dy = c.y[j] - c.y[i];
                const double v = std::sqrt(dx * dx + dy * dy) /
                                 std::abs(c.frame[j] - f);
                // the speed cap does not depend on the order: apply it
                // before the sort (same list as sorting first and filtering
                // after)
                if (has_vmax_ && !(v <= v_max_)) continue;
                s.cand.push_back(j);
                s.vel.push_back(v);
            }
            s.order.resize(s.cand.size());
            std::iota(s.order.begin(), s.order.end(), 0);
            std::stable_sort(s.order.begin(), s.order.end(),
                             [&](int p, int q) {
                                 return s.vel[p] < s.vel[q];
                             });
        }
        const size_t m = s.cand.size();
        // both branches met the candidates frame by frame, in ascending order
        size_t groups = 0;
        for (size_t k = 0; k < m; ++k)
            if (k == 0 || c.frame[s.cand[k]] != c.frame[s.cand[k - 1]])
                ++groups;
        // the grouped copy carries each tag's place in the seed queue with
        // it: half of the frontier candidates are dismissed on that alone
        // (seeds before the current one), without touching the tag's record
        int *out = m ? s.pool_alloc(3 * m + 2 * groups) : nullptr;
        for (size_t k = 0; k < m; ++k) out[k] = s.cand[(size_t)s.order[k]];
        int *grp = out + m, *count = nullptr;
        size_t w = 0;
        for (size_t k = 0; k < m; ++k) {
            const int j = s.cand[k];
            if (k == 0 || c.frame[j] != c.frame[s.cand[k - 1]]) {
                grp[w++] = c.frame[j];
                count = &grp[w++];
                *count = 0;
            }
            grp[w++] = j;
            grp[w++] = qpos_[(size_t)j];
            ++*count;
        }
        nb_ptr_[(size_t)i] = out;
        nb_len_[(size_t)i] = (int)m;
        grp_ptr_[(size_t)i] = grp;
        grp_len_[(size_t)i] = (int)w;
    }

    static void consult(int j, Scratch &s, SeedResult &res) {
        if (s.marks[(size_t)j].seen != s.run_stamp) {
            s.marks[(size_t)j].seen = s.run_stamp;
            s.consulted.push_back(j);
        }
    }

    // The frontier additions a member i brings: its neighbours.  Neither the
    // frontier nor the consulted set depends on the order they are visited
    // in, so without a speed cap (trial pass: every tag of the frames around
    // i qualifies) the two index ranges are walked directly -- no list, no
    // distances, no sort; the ordered list is only built for seeds.
    void add_neighbors_to_frontier(int i, int pos, Scratch &s,
                                   SeedResult &res) {
        if (has_vmax_) {
            // frames the arc already has a tag in are skipped as a whole: a
            // growing arc has taken most of the frames behind its new member
            neighbors(i, s);
            const int *p = grp_ptr_[(size_t)i];
            const int *const end = p + grp_len_[(size_t)i];
            while (p < end) {
                const int g = p[0], k = p[1];
                p += 2;
                if (s.taken[g] != s.stamp)
                    for (int q = 0; q < k; ++q)
                        if (p[2 * q + 1] >= pos)
                            add_listed_to_frontier(p[2 * q], g, s, res);
                p += 2 * k;
            }
            return;
        }
        const hk_clip &c = c_;
        const int f = c.frame[i];
        const int gap = c.p.max_frame_gap_in_arc;
        const int lo = std::max(0, f - gap - 1);
        const int hi = std::min(c.F, f + gap + 2);
        add_range_to_frontier(c.start[lo], c.start[f], pos, s, res);
        add_range_to_frontier(c.start[f + 1], c.start[hi], pos, s, res);
    }

    void add_range_to_frontier(int j0, int j1, int pos, Scratch &s,
                               SeedResult &res) {
        for (int j = j0; j < j1; ++j) add_one_to_frontier(j, pos, s, res);
    }

    // A tag of a grouped neighbour list: its frame g is not taken and its
    // queue position is not before the seed's (both checked by the caller).
    // Most such tags are in the frontier already -- a new member's neighbours
    // are largely those of the members before it -- and that is decided on
    // the thread's own marks, before the tag's record and its done flag
    // (shared with the other threads) are touched.  A tag in the frontier
    // was consulted when it was added.
    void add_listed_to_frontier(int j, int g, Scratch &s, SeedResult &res) {
        Scratch::Marks &m = s.marks[(size_t)j];
        if (m.in_frontier == s.stamp) return;
        if (done_[j].load(std::memory_order_relaxed)) return;
        if (m.seen != s.run_stamp) {
            m.seen = s.run_stamp;
            s.consulted.push_back(j);
        }
        m.in_frontier = s.stamp;
        const TagRec &t = tag_[(size_t)j];
        s.fr_push(j, (double)g, t.x, t.y);
    }

    void add_one_to_frontier(int j, int pos, Scratch &s, SeedResult &res) {
        {
            const TagRec &t = tag_[(size_t)j];
            if (t.qpos < pos || done_[j].load(std::memory_order_relaxed))
                return;
            Scratch::Marks &m = s.marks[(size_t)j];
            if (m.seen != s.run_stamp) {
                m.seen = s.run_stamp;
                s.consulted.push_back(j);
            }
            if (s.taken[t.frame] == s.stamp || m.in_frontier == s.stamp)
                return;
            m.in_frontier = s.stamp;
            s.fr_push(j, (double)t.frame, t.x, t.y);
        }
    }

    // One seed: try its neighbours in order until one pair grows into an
    // acceptable arc (reference tracker.py:521-625).
    // A trial arc can only be accepted if, when it stops growing, every
    // member lies within max_distance_pixels of it (the check after every
    // refit) and its e passes acceleration_ok (eval_arc).  Members are never
    // dropped, so if NO parabola y = c + e t^2 with e in the accepted window
    // passes within max_distance_pixels of three of the current members (in y
    // alone: with zero tilt a tag's distance to the arc is at least its
    // vertical offset), none will do so for any larger set either: the trial
    // is doomed and is given up at once -- same outcome (no arc), without the
    // growth steps in between, most of which belong to long trials along
    // slowly moving clutter whose fitted e is ~0.
    // For members at frames f1 < f2 < f3 the second divided difference of
    // their y is e + that of their offsets, so e lies within
    // D +- maxd * sum |coefficients|; likewise that of x is 0 +- the same.
    bool doomed(const Scratch &s) const {
        const size_t n = (size_t)s.member_count();
        if (!prune_ok_ || n < 3) return false;
        const size_t i1 = (size_t)s.m_lo, i2 = i1 + n / 2, i3 = i1 + n - 1;
        const double *const mf = s.mem_f.data();
        const double f1 = mf[i1], f2 = mf[i2], f3 = mf[i3];
        const double c1 = 1.0 / ((f2 - f1) * (f3 - f1));
        const double c2 = 1.0 / ((f2 - f1) * (f3 - f2));
        const double c3 = 1.0 / ((f3 - f1) * (f3 - f2));
        if (!(c1 > 0 && c2 > 0 && c3 > 0)) return false;   // frames not distinct
        const double slack = maxd_ * (c1 + c2 + c3) * (1 + 1e-9) + 1e-12;
        const double dy = s.mem_y[i1] * c1 - s.mem_y[i2] * c2 + s.mem_y[i3] * c3;
        const double dx = s.mem_x[i1] * c1 - s.mem_x[i2] * c2 + s.mem_x[i3] * c3;
        if (!(dy == dy) || !(dx == dx)) return false;       // NaN coordinates
        const double tol = 1e-9 * (std::fabs(dy) + std::fabs(dx) + 1.0);
        if (dy + slack + tol < e_lo_) return true;
        if (dy - slack - tol > e_hi_) return true;
        if (std::fabs(dx) - slack - tol > 0) return true;
        return false;
    }

    // Squared distance of a tag at (frame f, x, y) to the arc: the operations
    // of arc_tag_distance, minus the final square root.
    double dist_sq(const Arc &arc, double f, double x, double y) const {
        const double t = f - arc.f_peak;
        const double xj = arc.a + arc.b * t;
        const double yj = arc.c + arc.e * t * t;
        const double px = xj * rot_.c + yj * rot_.s;
        const double py = yj * rot_.c - xj * rot_.s;
        const double dx = px - x, dy = py - y;
        return dx * dx + dy * dy;
    }

    // sqrt(q) > maxd_, deciding without the root unless q is within rounding
    // of the boundary (sqrt is monotonic and correctly rounded).
    bool farther_than_maxd(double q) const {
        if (q > maxd_sq_hi_) return true;
        if (q < maxd_sq_lo_) return false;
        return std::sqrt(q) > maxd_;
    }

    // fit_one() for unit weights over the member arrays (1.0 * v == v, so
    // the weight factors drop out bit for bit).
    void fit_members(Arc &arc, const Scratch &s) const {
        const size_t n = (size_t)s.member_count();
        if (n < 3) return;
        Moments m;
        const double fp = arc.f_peak;
        const double *const mf = s.mem_f.data() + s.m_lo;
        const double *const mx = s.mem_x.data() + s.m_lo;
        const double *const my = s.mem_y.data() + s.m_lo;
        if (simd_) {
            double v[11];
            simd_moments(mf, mx, my, n, fp, v);
            m.T0 = v[0]; m.T1 = v[1]; m.T2 = v[2]; m.T3 = v[3]; m.T4 = v[4];
            m.X1 = v[5]; m.TX = v[6]; m.T2X = v[7];
            m.Y1 = v[8]; m.TY = v[9]; m.T2Y = v[10];
            solve_fit(arc, m, rot_);
            return;
        }
        for (size_t i = 0; i < n; ++i) {
            const double t = mf[i] - fp;
            const double x = mx[i], y = my[i];
            const double t2 = t * t;
            m.T0 += 1.0;
            m.T1 += t;
            m.T2 += t2;
            m.T3 += t2 * t;
            m.T4 += t2 * t2;
            m.X1 += x;
            m.TX += t * x;
            m.T2X += t2 * x;
            m.Y1 += y;
            m.TY += t * y;
            m.T2Y += t2 * y;
        }
        solve_fit(arc, m, rot_);
    }

    // One pass over the frontier: drops the tags of frame `dead` (the frame
    // the last new member took -- the only one that became taken; order is
    // kept), computes every remaining tag's squared distance to the arc and
    // finds the nearest: smallest distance, lowest tag index among equals.
    // Compared on squared distances; roots are only taken among those within
    // rounding of the smallest.  Returns the number of tags left, with the
    // nearest one's position in `at` and its distance in `best_d`.  The first
    // `known` tags have their squared distance to this arc in fr_q already
    // (the arc has not been refitted since the last scan).
    size_t scan_frontier(Scratch &s, const Arc &arc, double dead, size_t known,
                         size_t &at, double &best_d) const {
        const size_t nf = s.nf;
        int *const id = s.fr_id.data();
        double *const ff = s.fr_f.data(), *const fx = s.fr_x.data(),
                     *const fy = s.fr_y.data(), *const fq = s.fr_q.data();
        const double pa = arc.a, pb = arc.b, pc = arc.c, pe = arc.e,
                     pf = arc.f_peak, rs = rot_.s, rc = rot_.c;
        const double inf = std::numeric_limits<double>::infinity();
        double q1 = inf, q2 = inf;      // smallest and second smallest
        size_t i1 = 0, k = 0;
        bool plain = true;              // no NaN met
        for (size_t i = 0; i < nf; ++i) {
            const double f = ff[i];
            if (f == dead) continue;
            const double x = fx[i], y = fy[i];
            double q;
            if (i < known) {
                q = fq[i];
            } else {    // the operations of dist_sq(), in its order
                const double t = f - pf;
                const double xj = pa + pb * t;
                const double yj = pc + pe * t * t;
                const double px = xj * rc + yj * rs;
                const double py = yj * rc - xj * rs;
                const double dx = px - x, dy = py - y;
                q = dx * dx + dy * dy;
            }
            if (k != i) {
                id[k] = id[i];
                ff[k] = f;
                fx[k] = x;
                fy[k] = y;
            }
            fq[k] = q;
            if (q < q1) {
                q2 = q1;
                q1 = q;
                i1 = k;
            } else if (q < q2) {
                q2 = q;
            } else if (q != q) {
                plain = false;
            }
            ++k;
        }
        s.nf = k;
        if (k == 0) return 0;
        // nobody else within rounding of the smallest (nearly always)
        if (plain && q2 > q1 * (1 + 1e-14)) {
            at = i1;
            best_d = std::sqrt(q1);
            return k;
        }
        double q_min = 0;
        for (size_t i = 0; i < k; ++i)
            if (i == 0 || fq[i] < q_min) q_min = fq[i];
        int best = -1;
        best_d = 0;
        if (q_min == q_min) {
            const double q_tie = q_min * (1 + 1e-14);
            for (size_t i = 0; i < k; ++i) {
                if (!(fq[i] <= q_tie)) continue;
                const double d = std::sqrt(fq[i]);
                if (best < 0 || d < best_d || (d == best_d && id[i] < best)) {
                    best = id[i];
                    best_d = d;
                    at = i;
                }
            }
        }
        if (best < 0) {         // NaN distances: the literal scan
            for (size_t i = 0; i < k; ++i) {
                const double d = std::sqrt(fq[i]);
                if (best < 0 || d < best_d || (d == best_d && id[i] < best)) {
                    best = id[i];
                    best_d = d;
                    at = i;
                }
            }
        }
        return k;
    }

    // One seed: try its neighbours in order until one pair grows into an
    // acceptable arc (reference tracker.py:521-625).
    void run_seed(int t1, int pos, Scratch &s, SeedResult &res) {
        s.consulted.clear();
        grow_seed(t1, pos, s, res);
        res.consulted.assign(s.consulted.begin(), s.consulted.end());
    }

    void grow_seed(int t1, int pos, Scratch &s, SeedResult &res) {
        const hk_clip &c = c_;
        res.made = false;
        ++s.run_stamp;
        ++s.runs;
        Span nb1;
        S3P_SCOPE(11);
        { S3P_T0(5); nb1 = neighbors(t1, s); S3P_ADD(5); }
        const int f1 = c.frame[t1];
        const double x1 = c.x[t1], y1 = c.y[t1];
        for (int t2 : nb1) {
            if (is_done(t2, pos)) continue;
            consult(t2, s, res);
            S3P_SCOPE(10);
            ++s.stamp;
            Arc arc;
            const int f2 = c.frame[t2];
            const double x2 = c.x[t2], y2 = c.y[t2];
            s.members_clear();
            if (t1 < t2) {
                s.member_insert(t1, (double)f1, x1, y1);
                s.member_insert(t2, (double)f2, x2, y2);
            } else {
                s.member_insert(t2, (double)f2, x2, y2);
                s.member_insert(t1, (double)f1, x1, y1);
            }
            s.taken[f1] = s.stamp;
            s.taken[f2] = s.stamp;
            s.nf = 0;
            { S3P_T0(0);
            add_neighbors_to_frontier(t1, pos, s, res);
            add_neighbors_to_frontier(t2, pos, s, res);
            S3P_ADD(0); }

            arc.e = e0_;
            arc.f_peak = (y1 - y2) / (2.0 * arc.e * (f2 - f1)) +
                         0.5 * (f1 + f2);
            const double dt1 = f1 - arc.f_peak;
            arc.c = y1 - arc.e * (dt1 * dt1);
            arc.b = (x1 - x2) / (f1 - f2);
            arc.a = x1 - arc.b * (f1 - arc.f_peak);

            bool bad = false;
            double dead = -1.0;     // no frame: nothing to drop yet
            size_t known = 0;       // frontier tags whose fr_q is up to date
            while (s.nf > 0) {
                size_t at = 0;
                double best_d = 0;
                S3P_CNT(6, s.nf);
                {
                    S3P_T0(1);
                    const size_t left =
                        scan_frontier(s, arc, dead, known, at, best_d);
                    S3P_ADD(1);
                    if (left == 0) break;
                    known = left;
                }
                // an arc committed meanwhile took this seed: the committer
                // will skip it, whatever this run would find
                if (done_[t1].load(std::memory_order_relaxed)) return;
                if (best_d > maxd_) break;
                ++s.steps;
                const int best = s.fr_id[at];
                const double fb = s.fr_f[at];       // its frame
                bool is_doomed;
                {
                    S3P_SCOPE(8);
                    s.member_insert(best, fb, s.fr_x[at], s.fr_y[at]);
                    s.taken[(size_t)(int)fb] = s.stamp;
                    is_doomed = doomed(s);
                }
                if (is_doomed) {
                    S3P_END(1, s.member_count());
                    bad = true;
                    break;
                }
                { S3P_T0(2);
                add_neighbors_to_frontier(best, pos, s, res);
                S3P_ADD(2); }
                dead = fb;      // dropped by the next scan
                bool refit = false;
                if (s.member_count() > c.p.min_tags_to_curve_fit) {
                    S3P_T0(3);
                    fit_members(arc, s);
                    S3P_ADD(3);
                    refit = true;
                    known = 0;
                }
                if (std::isnan(arc.e) || arc.e <= 0) {
                    S3P_END(2, s.member_count());
                    bad = true;
                    break;
                }
                if (refit) {
                    S3P_T0(4);
                    const size_t nm = (size_t)s.member_count();
                    const double *const mf = s.mem_f.data() + s.m_lo;
                    const double *const mx = s.mem_x.data() + s.m_lo;
                    const double *const my = s.mem_y.data() + s.m_lo;
                    if (simd_) {
                        const double p5[5] = {arc.a, arc.b, arc.c, arc.e,
                                              arc.f_peak};
                        bad = simd_any_farther(p5, rot_.s, rot_.c, mf, mx, my,
                                               nm, maxd_, maxd_sq_lo_,
                                               maxd_sq_hi_);
                    } else {
                        for (size_t i = 0; i < nm; ++i)
                            if (farther_than_maxd(
                                    dist_sq(arc, mf[i], mx[i], my[i]))) {
                                bad = true;
                                break;
                            }
                    }
                    S3P_ADD(4);
                }
                // without a refit the arc is unchanged: the older members
                // passed this test already and the new one has
                // best_d <= maxd
                if (bad) {
                    S3P_END(3, s.member_count());
                    break;
                }
            }
            // eval_arc() fails with fewer than min_tags_per_arc members
            // whatever else holds: skip the call
            if (!bad && s.member_count() >= c.p.min_tags_per_arc) {
                arc.tags.assign(s.mem_id.data() + s.m_lo,
                                s.mem_id.data() + s.m_hi);
                if (eval_arc_host(c, arc, rot_, false, true) == 0) {
                    S3P_END(6, s.member_count());
                    res.arc = std::move(arc);
                    res.made = true;
                    return;
                }
                S3P_END(5, s.member_count());
            } else if (!bad) {
                S3P_END(s.member_count() == 2 ? 0 : 4, s.member_count());
            }
        }
    }

    struct TagRec {                 // what the growth loop reads per tag
        double x, y;
        int frame, qpos;
    };
    alignas(128) std::vector<TagRec> tag_;

    hk_clip &c_;
    const int n_;
    SeedDevice *dev_ = nullptr;         // full pass only
    int dev_window_ = 0;
    std::unique_ptr<DevRound[]> dev_rounds_;
    std::atomic<int> dev_error_{0};
    char dev_message_[256] = {0};
    std::vector<int> queue_, qpos_;     // seeds top of the frame first
    std::vector<int> xs_idx_;           // tags by (frame, x)
    std::vector<double> xs_val_;        // their x
    bool x_index_ok_ = false;
    std::vector<const int *> nb_ptr_;
    std::vector<int> nb_len_;
    // the same neighbours grouped by frame: {frame, count, tags...} per
    // frame that has any (what the frontier is fed from)
    std::vector<const int *> grp_ptr_;
    std::vector<int> grp_len_;
    std::unique_ptr<std::atomic<uint8_t>[]> nb_state_;
    std::unique_ptr<std::atomic<uint8_t>[]> done_;
    std::unique_ptr<std::atomic<uint8_t>[]> taken_;  // member of an accepted arc
    std::unique_ptr<Scratch[]> scratch_;
    std::unique_ptr<Slot[]> slots_;
    size_t window_ = 1;
    std::vector<long> done_at_;     // committer only: accept number of the
    size_t n_accepted_ = 0;         // arc that took the tag, or -1
    std::vector<std::thread> pool_;
    alignas(128) std::atomic<size_t> next_{0};
    alignas(128) std::atomic<int> build_next_{0};   // tags whose lists are built
    alignas(128) std::atomic<size_t> commit_pos_{0};
    alignas(128) std::atomic<size_t> accepted_{0};
    std::atomic<size_t> accepting_{0};  // accepted_, raised before the flags
    alignas(128) std::atomic<int> active_{0};
    std::atomic<unsigned> generation_{0};
    std::atomic<bool> quit_{false}, stop_{false};
    int nthreads_ = 1;
    unsigned helper_slots_used_ = 0;    // scratch slots of helpers (g_open_mu)
    // HK_STEP3_SIMD=0: the scalar loops (tests compare the two)
    const bool simd_ = simd_available() &&
                       !(getenv("HK_STEP3_SIMD") &&
                         atoi(getenv("HK_STEP3_SIMD")) == 0);
    bool has_vmax_ = false;
    double v_max_ = 0, maxd_ = 0, maxd_sq_lo_ = 0, maxd_sq_hi_ = 0, e0_ = 0;
    double e_lo_ = 0, e_hi_ = 0;        // window of acceleration_ok, see doomed()
    bool prune_ok_ = false;
    Rot rot_{0, 1};
};

void find_global_params(hk_clip &c, const std::vector<Arc> &arcs) {
    if (arcs.empty()) return;
    std::vector<const Arc *> top;
    top.reserve(arcs.size());
    for (const Arc &a : arcs) top.push_back(&a);
    std::stable_sort(top.begin(), top.end(), [](const Arc *p, const Arc *q) {
        return p->tags.size() > q->tags.size();
    });
    if (top.size() > 10) top.resize(10);
    std::vector<double> es;
    for (const Arc *a : top) es.push_back(a->e);
    std::sort(es.begin(), es.end());
    const size_t n = es.size();
    const double med = (n % 2) ? es[n / 2] : (es[n / 2 - 1] + es[n / 2]) / 2;
    c.g = 2 * med;
    c.has_g = true;
    c.cm_per_pixel = G_CM / (c.g * (c.fps * c.fps));
}

// ---------------------------------------------------------------------------
// step 4 host parts
// ---------------------------------------------------------------------------

struct MergeCache {         // per arc, valid while the arc is unchanged
    bool valid = false;
    int f_min = 0, f_max = 0;
    std::vector<int> good;
};

void refresh_cache(const hk_clip &c, const Arc &arc, const Rot &rot,
                   const double *dist, MergeCache &mc) {
    // dist: member distances (arc.tags order) from K7, or null -> host
    mc.valid = true;
    mc.good.clear();
    if (arc.tags.empty()) return;
    arc_frame_range(c, arc, mc.f_min, mc.f_max);
    const double m = median_tag_radius(c, arc.tags);
    const double r_lo = m * (1 - c.p.radius_window);
    const double r_hi = m * (1 + c.p.radius_window);
    for (size_t k = 0; k < arc.tags.size(); ++k) {
        const int i = arc.tags[k];
        const double d = dist ? dist[k] : arc_tag_distance(c, arc, rot, i);
        if (d < c.p.max_distance_pixels && c.tw[i] > 0.95 && r_lo <= c.r[i] &&
            c.r[i] <= r_hi)
            mc.good.push_back(i);
    }
}

// One merge_arcs() trial on the host: union of the two good-tag lists, fit,
// outlier count (reference tracker.py:1199-1233).
void host_merge_trial(const hk_clip &c, const Arc &a1, const MergeCache &m1,
                      const MergeCache &m2, const Rot &rot,
                      std::vector<int> &trial_tags, MergeTrial &out) {
    trial_tags.clear();
    std::set_union(m1.good.begin(), m1.good.end(), m2.good.begin(),
                   m2.good.end(), std::back_inserter(trial_tags));
    out.p[0] = a1.a;            // copy.copy(arc1): parameters only
    out.p[1] = a1.b;
    out.p[2] = a1.c;
    out.p[3] = a1.e;
    out.p[4] = a1.f_peak;
    out.union_n = (int)trial_tags.size();
    out.poor = 0;
    if (out.union_n < c.p.min_tags_per_arc) return;
    if (out.union_n >= 3) {
        Moments m;
        const double fp = out.p[4];
        for (int i : trial_tags)
            moments_add(m, c.frame[i] - fp, c.tw[i], c.x[i], c.y[i]);
        solve_fit5(out.p, m, rot.s, rot.c);
    }
    const double maxd = c.p.max_distance_pixels;
    for (int i : trial_tags)
        if (std::sqrt(arc_dist_sq5(out.p, (double)c.frame[i], c.x[i], c.y[i],
                                   rot.s, rot.c)) > maxd &&
            ++out.poor > 2)
            break;
}

// All of "while self.merge_arcs(arcs): ..." (reference tracker.py:964-965,
// 1126-1286).  Decisions and their order are the reference's; the work is
// organised differently:
//  * the good-tag lists and frame ranges the reference recomputes for every
//    pair are cached per arc and refreshed only for the arc a merger changed;
//  * the trial fits of ALL candidate pairs of the pass are evaluated up
//    front in one kernel launch (K8 merge_trials_k, bit-identical to the
//    host's trial fit) -- for all clips of a batch at once, which is why the
//    phase comes in two halves: merge_prepare() lists a clip's candidate
//    pairs, em_merge_trials() runs those of every clip, and merge_scan() then
//    only looks results up, computing on the host the pairs that involve an
//    arc changed by a merger of this pass.
struct MergeWork {
    Rot rot{0, 1};
    std::vector<MergeCache> cache;
    // member distances per arc (arc.tags order): from K7 for the arcs as the
    // pass found them, recomputed here for an arc a merger changed; what
    // prune_phase() then reads instead of asking the device again
    std::vector<std::vector<double>> adist;
    MergeInput in;                  // good-tag lists (CSR) and pairs
    std::vector<int> pair_start;    // first pair of each i1
    bool launched = false;          // the pairs went to the device
};

void merge_prepare(hk_clip &c, const std::vector<double> &dist,
                   const std::vector<long> &off, MergeWork &mw) {
    mw.rot = rotation(c);
    const size_t A0 = c.arcs.size();
    mw.cache.assign(A0, MergeCache());
    mw.adist.assign(A0, std::vector<double>());
    for (size_t k = 0; k < A0; ++k) {
        refresh_cache(c, c.arcs[k], mw.rot, dist.data() + off[k], mw.cache[k]);
        mw.adist[k].assign(dist.begin() + off[k], dist.begin() + off[k + 1]);
    }
    const std::vector<MergeCache> &cache = mw.cache;
    auto candidate = [&](size_t i1, size_t i2) {    // the cheap tests
        return !(cache[i2].f_max < cache[i1].f_min ||
                 cache[i2].f_min > cache[i1].f_max) &&
               cache[i2].good.size() >= 3;
    };
    // every pair the scan could reach with the arcs as they are now
    // (pairs are generated by ascending i1, then ascending i2: pair_start[i1]
    // = first pair of i1, and i2 is found by bisection -- memory stays linear
    // in the number of candidate pairs however many arcs there are)
    std::vector<int> &pairs = mw.in.pairs;
    pairs.clear();
    mw.in.good_off.clear();
    mw.in.good.clear();
    mw.pair_start.clear();
    mw.launched = false;
    for (size_t i1 = 0; i1 < A0; ++i1) {
        if (c.arcs[i1].done || c.arcs[i1].tags.empty() ||
            cache[i1].good.size() < 3)
            continue;
        for (size_t i2 = 0; i2 < A0; ++i2)
            if (i2 != i1 && !c.arcs[i2].tags.empty() && candidate(i1, i2)) {
                pairs.push_back((int)i1);
                pairs.push_back((int)i2);
            }
    }
    if (pairs.size() / 2 >= 128) {      // else not worth a place in the launch
        mw.in.good_off.assign(A0 + 1, 0);
        for (size_t k = 0; k < A0; ++k) {
            mw.in.good.insert(mw.in.good.end(), cache[k].good.begin(),
                              cache[k].good.end());
            mw.in.good_off[k + 1] = (int)mw.in.good.size();
        }
        mw.pair_start.assign(A0 + 1, 0);
        for (size_t q = 0; q < pairs.size() / 2; ++q)
            ++mw.pair_start[(size_t)pairs[2 * q] + 1];
        for (size_t k = 0; k < A0; ++k)
            mw.pair_start[k + 1] += mw.pair_start[k];
        mw.launched = true;
    } else {
        pairs.clear();
    }
}

int merge_scan(hk_clip &c, MergeWork &mw, const std::vector<MergeTrial> &trials,
               bool &merged_any) {
    const Rot rot = mw.rot;
    std::vector<MergeCache> &cache = mw.cache;
    const std::vector<int> &pairs = mw.in.pairs;
    const std::vector<int> &pair_start = mw.pair_start;
    const size_t A0 = cache.size();
    const bool verify = g_self_check.load() != 0;
    auto candidate = [&](size_t i1, size_t i2) {
        return !(cache[i2].f_max < cache[i1].f_min ||
                 cache[i2].f_min > cache[i1].f_max) &&
               cache[i2].good.size() >= 3;
    };
    auto find_trial = [&](size_t o1, size_t o2) -> const MergeTrial * {
        if (!mw.launched) return nullptr;
        int lo = pair_start[o1], hi = pair_start[o1 + 1];
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (pairs[2 * (size_t)mid + 1] < (int)o2)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < pair_start[o1 + 1] && pairs[2 * (size_t)lo + 1] == (int)o2)
            return &trials[(size_t)lo];
        return nullptr;
    };
    std::vector<int> orig(A0);              // position now -> position then
    std::iota(orig.begin(), orig.end(), 0);
    std::vector<char> changed(A0, 0);

    std::vector<int> trial_tags;
    MergeTrial host_trial;
    bool merged = true;
    while (merged) {                    // one iteration = one merge_arcs()
        merged = false;
        for (size_t i1 = 0; i1 < c.arcs.size() && !merged; ++i1) {
            Arc &a1 = c.arcs[i1];
            if (a1.done) continue;
            if (a1.tags.empty() || cache[i1].good.size() < 3) {
                a1.done = true;
                continue;
            }
            for (size_t i2 = 0; i2 < c.arcs.size(); ++i2) {
                if (i2 == i1) continue;
                Arc &a2 = c.arcs[i2];
                if (a2.tags.empty()) {
                    a2.done = true;
                    continue;
                }
                if (!candidate(i1, i2)) continue;
                const MergeTrial *tr = nullptr;
                const size_t o1 = (size_t)orig[i1], o2 = (size_t)orig[i2];
                if (!changed[o1] && !changed[o2]) tr = find_trial(o1, o2);
                if (tr == nullptr || verify) {
                    host_merge_trial(c, a1, cache[i1], cache[i2], rot,
                                     trial_tags, host_trial);
                    if (tr != nullptr &&
                        (tr->union_n != host_trial.union_n ||
                         tr->poor != host_trial.poor ||
                         std::memcmp(tr->p, host_trial.p, sizeof(tr->p)) != 0))
                        return fail(HK_ERR_INTERNAL,
                                    "merge trial of arcs %zu, %zu differs "
                                    "between device and host", i1, i2);
                    tr = &host_trial;
                }
                if (tr->union_n < c.p.min_tags_per_arc) continue;
                if (tr->poor > 2) continue;

                // merge a2 into a1
                if (tr != &host_trial) {
                    trial_tags.clear();
                    std::set_union(cache[i1].good.begin(),
                                   cache[i1].good.end(),
                                   cache[i2].good.begin(),
                                   cache[i2].good.end(),
                                   std::back_inserter(trial_tags));
                }
                a1.a = tr->p[0];
                a1.b = tr->p[1];
                a1.c = tr->p[2];
                a1.e = tr->p[3];
                a1.f_peak = tr->p[4];
                a1.tags = trial_tags;
                a1.w_off = -1;
                refresh_cache(c, a1, rot, nullptr, cache[i1]);
                mw.adist[i1].resize(a1.tags.size());
                for (size_t k = 0; k < a1.tags.size(); ++k)
                    mw.adist[i1][k] = arc_tag_distance(c, a1, rot, a1.tags[k]);
                changed[o1] = 1;
                c.arcs.erase(c.arcs.begin() + (long)i2);
                cache.erase(cache.begin() + (long)i2);
                mw.adist.erase(mw.adist.begin() + (long)i2);
                orig.erase(orig.begin() + (long)i2);
                merged = true;
                merged_any = true;
                break;
            }
            if (!merged) {
                c.arcs[i1].done = true;
                const int rc = report_progress(c);
                if (rc != HK_OK) return rc;
            }
        }
    }
    return HK_OK;
}

// All of "while self.prune_arcs(arcs): ..." (reference tracker.py:969-970,
// 1288-1329): arcs do not change between passes, so every failing arc goes.
// adist: the member distances of every arc as it stands (see MergeWork).
int prune_phase(hk_clip &c, const std::vector<std::vector<double>> &adist,
                bool &pruned_any) {
    std::vector<Arc> kept;
    kept.reserve(c.arcs.size());
    for (size_t k = 0; k < c.arcs.size(); ++k) {
        const Arc &arc = c.arcs[k];
        int code = 0;
        if (std::isnan(arc.a) || std::isnan(arc.b) || std::isnan(arc.c) ||
            std::isnan(arc.e) || std::isnan(arc.f_peak)) {
            code = 1;
        } else if (arc.tags.empty()) {
            code = 5;
        } else if (!acceleration_ok(c, arc.e)) {
            code = 4;
        } else {
            int close = 0;
            for (double dd : adist[k])
                if (dd < c.p.max_distance_pixels) ++close;
            if (close < c.p.min_tags_per_arc) code = 2;
        }
        if (code == 0) {
            kept.push_back(arc);
        } else {
            pruned_any = true;
            const int prc = report_progress(c);
            if (prc != HK_OK) return prc;
        }
    }
    c.arcs.swap(kept);
    return HK_OK;
}

// clean_notes (reference tracker.py:1331-1422).
int clean(hk_clip &c) {
    const double maxd = c.p.max_distance_pixels;
    const Rot rot = rotation(c);
    const size_t A = c.arcs.size();
    // tag.weight keys, in arc-list order: per tag the arcs that hold it (one
    // flat list with offsets; most tags are on no arc at all)
    std::vector<int> ta_off((size_t)c.n + 1, 0), ta;
    for (size_t k = 0; k < A; ++k)
        for (int i : c.arcs[k].tags) ++ta_off[(size_t)i + 1];
    for (int i = 0; i < c.n; ++i) ta_off[(size_t)i + 1] += ta_off[(size_t)i];
    ta.resize((size_t)ta_off[(size_t)c.n]);
    {
        std::vector<int> at(ta_off.begin(), ta_off.end() - 1);
        for (size_t k = 0; k < A; ++k)
            for (int i : c.arcs[k].tags) ta[(size_t)at[(size_t)i]++] = (int)k;
    }
    struct ArcsOf {
        const int *b, *e;
        const int *begin() const { return b; }
        const int *end() const { return e; }
    };
    auto tag_arcs = [&](int i) {
        return ArcsOf{ta.data() + ta_off[(size_t)i],
                      ta.data() + ta_off[(size_t)i + 1]};
    };
    std::vector<char> tag_done(c.n, 0), arc_dead(A, 0), arc_done(A, 0);

    bool again = true;
    while (again) {
        for (int i = 0; i < c.n; ++i) {
            if (!c.alive[i] || tag_done[i]) continue;
            bool good = false;
            for (int k : tag_arcs(i))
                if (!arc_dead[k] &&
                    arc_tag_distance(c, c.arcs[k], rot, i) < maxd) {
                    good = true;
                    break;
                }
            if (good) {
                tag_done[i] = 1;
                continue;
            }
            c.alive[i] = 0;
            for (int k : tag_arcs(i)) {
                if (arc_dead[k]) continue;
                std::vector<int> &tg = c.arcs[k].tags;
                const auto it = std::lower_bound(tg.begin(), tg.end(), i);
                if (it != tg.end() && *it == i) tg.erase(it);
                arc_done[k] = 0;
            }
        }
        const int rc = report_progress(c);
        if (rc != HK_OK) return rc;
        again = false;
        for (size_t k = 0; k < A; ++k) {
            if (arc_dead[k] || arc_done[k]) continue;
            if (eval_arc_host(c, c.arcs[k], rot, true, true) > 0) {
                arc_dead[k] = 1;
                for (int i : c.arcs[k].tags) tag_done[i] = 0;
                again = true;
            } else {
                arc_done[k] = 1;
            }
        }
    }

    // final hard assignment: nearest arc among the tag's weight keys
    std::vector<int> new_index(A, -1);
    std::vector<Arc> kept;
    for (size_t k = 0; k < A; ++k) {
        if (arc_dead[k]) continue;
        new_index[k] = (int)kept.size();
        kept.push_back(c.arcs[k]);
        kept.back().tags.clear();
        kept.back().w_off = -1;
        kept.back().done = false;
    }
    std::fill(c.final_arc.begin(), c.final_arc.end(), -1);
    for (int i = 0; i < c.n; ++i) {
        if (!c.alive[i]) continue;
        int best = -1;
        double best_d = 0;
        for (int k : tag_arcs(i)) {
            if (arc_dead[k]) continue;
            const double d = arc_tag_distance(c, c.arcs[k], rot, i);
            if (best < 0 || d < best_d) {
                best = k;
                best_d = d;
            }
        }
        if (best < 0)
            return fail(HK_ERR_INTERNAL, "clean: surviving tag without arc");
        c.final_arc[i] = new_index[best];
        kept[new_index[best]].tags.push_back(i);
    }
    c.arcs.swap(kept);
    return HK_OK;
}

void sort_arcs_by_peak(hk_clip &c) {
    const size_t A = c.arcs.size();
    std::vector<int> order(A);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) {
        return c.arcs[p].f_peak < c.arcs[q].f_peak;
    });
    std::vector<int> where(A);
    std::vector<Arc> sorted;
    sorted.reserve(A);
    for (size_t k = 0; k < A; ++k) {
        where[order[k]] = (int)k;
        sorted.push_back(std::move(c.arcs[order[k]]));
    }
    c.arcs.swap(sorted);
    for (int &fa : c.final_arc)
        if (fa >= 0) fa = where[fa];
}

}  // namespace

}  // namespace hk

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

using hk::Arc;

extern "C" int hk_scan_params_default(hk_scan_params *p, int frame_height,
                                      int scan_height, double fps) {
    if (!p || frame_height <= 0 || scan_height <= 0)
        return hk::fail(HK_ERR_ARG, "hk_scan_params_default: bad argument");
    p->sigmasq = 15.0;
    p->max_frame_gap_in_arc = 2;
    p->g_window = 0.25;
    p->min_tags_to_curve_fit = 4;
    p->default_frame_height_cm = 300;
    p->max_frame_height_cm = 1000;
    p->default_fps = 30;
    p->max_fps = 60;
    p->min_tags_per_arc = fps >= 29 ? 10 : 5;
    p->max_distance_pixels = 5.0 * frame_height / 480;
    p->radius_window = scan_height >= 480 ? 0.65 : 0.75;
    return HK_OK;
}

extern "C" int hk_clip_create(int frame_width, int frame_height, double fps,
                              int frame_count, const hk_scan_params *params,
                              int ntags, const int *frame, const double *x,
                              const double *y, const double *radius,
                              const double *total_weight, int device,
                              hk_clip **out) {
    if (!out || !params || frame_width <= 0 || frame_height <= 0 ||
        frame_count < 0 || ntags < 0 || !(fps > 0))
        return hk::fail(HK_ERR_ARG, "hk_clip_create: bad argument");
    if (ntags > 0 && (!frame || !x || !y || !radius || !total_weight))
        return hk::fail(HK_ERR_ARG, "hk_clip_create: null event array");
    for (int i = 0; i < ntags; ++i) {
        if (frame[i] < 0 || frame[i] >= frame_count ||
            (i > 0 && frame[i] < frame[i - 1]))
            return hk::fail(HK_ERR_ARG,
                            "hk_clip_create: frame[] must be ascending within "
                            "[0, frame_count); entry %d is %d", i, frame[i]);
    }
    hk_clip *c = new hk_clip();
    c->W = frame_width;
    c->H = frame_height;
    c->F = frame_count;
    c->fps = fps;
    c->p = *params;
    c->n = ntags;
    c->frame.assign(frame, frame + ntags);
    c->x.assign(x, x + ntags);
    c->y.assign(y, y + ntags);
    c->r.assign(radius, radius + ntags);
    c->tw.assign(total_weight, total_weight + ntags);
    c->start.assign((size_t)frame_count + 1, 0);
    for (int i = 0; i < ntags; ++i) ++c->start[(size_t)frame[i] + 1];
    for (int f = 0; f < frame_count; ++f) c->start[f + 1] += c->start[f];
    c->alive.assign(ntags, 1);
    c->final_arc.assign(ntags, -1);
    c->device = device;
    *out = c;
    return HK_OK;
}

extern "C" int hk_clip_destroy(hk_clip *c) {
    if (!c) return HK_OK;
    hk::em_release(*c);
    delete c;
    return HK_OK;
}

extern "C" int hk_clip_set_progress(hk_clip *c, hk_progress_fn fn,
                                    void *user) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_set_progress: null clip");
    c->progress = fn;
    c->progress_user = user;
    return HK_OK;
}

extern "C" int hk_clip_set_state(hk_clip *c, double camera_tilt, int has_g,
                                 double g) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_set_state: null clip");
    c->tilt = camera_tilt;
    c->has_g = has_g != 0;
    c->g = has_g ? g : 0.0;
    if (has_g) c->cm_per_pixel = 980.7 / (g * (c->fps * c->fps));
    return HK_OK;
}

extern "C" int hk_clip_get_state(const hk_clip *c, double *camera_tilt,
                                 int *has_g, double *g, double *cm_per_pixel) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_get_state: null clip");
    if (camera_tilt) *camera_tilt = c->tilt;
    if (has_g) *has_g = c->has_g ? 1 : 0;
    if (g) *g = c->g;
    if (cm_per_pixel) *cm_per_pixel = c->cm_per_pixel;
    return HK_OK;
}

extern "C" int hk_clip_set_arcs(hk_clip *c, int narcs, const double *params5,
                                const int *ids, const int *tag_offsets,
                                const int *tag_index) {
    if (!c || narcs < 0 || (narcs > 0 && (!params5 || !tag_offsets)))
        return hk::fail(HK_ERR_ARG, "hk_clip_set_arcs: bad argument");
    std::vector<Arc> arcs((size_t)narcs);
    for (int k = 0; k < narcs; ++k) {
        Arc &a = arcs[k];
        a.a = params5[5 * k];
        a.b = params5[5 * k + 1];
        a.c = params5[5 * k + 2];
        a.e = params5[5 * k + 3];
        a.f_peak = params5[5 * k + 4];
        a.id = ids ? ids[k] : k + 1;
        const int lo = tag_offsets[k], hi = tag_offsets[k + 1];
        if (lo < 0 || hi < lo || (hi > lo && !tag_index))
            return hk::fail(HK_ERR_ARG, "hk_clip_set_arcs: bad offsets");
        a.tags.assign(tag_index + lo, tag_index + hi);
        for (int i : a.tags)
            if (i < 0 || i >= c->n)
                return hk::fail(HK_ERR_ARG,
                                "hk_clip_set_arcs: tag index %d out of range",
                                i);
        std::sort(a.tags.begin(), a.tags.end());
        a.tags.erase(std::unique(a.tags.begin(), a.tags.end()), a.tags.end());
    }
    c->arcs.swap(arcs);
    return HK_OK;
}

extern "C" int hk_clip_arc_count(const hk_clip *c) {
    return c ? (int)c->arcs.size() : 0;
}

extern "C" int hk_clip_arc_tag_total(const hk_clip *c) {
    if (!c) return 0;
    size_t n = 0;
    for (const Arc &a : c->arcs) n += a.tags.size();
    return (int)n;
}

extern "C" int hk_clip_get_arcs(const hk_clip *c, double *params5, int *ids,
                                int *tag_offsets, int *tag_index) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_get_arcs: null clip");
    int pos = 0;
    for (size_t k = 0; k < c->arcs.size(); ++k) {
        const Arc &a = c->arcs[k];
        if (params5) {
            params5[5 * k] = a.a;
            params5[5 * k + 1] = a.b;
            params5[5 * k + 2] = a.c;
            params5[5 * k + 3] = a.e;
            params5[5 * k + 4] = a.f_peak;
        }
        if (ids) ids[k] = a.id;
        if (tag_offsets) tag_offsets[k] = pos;
        if (tag_index)
            std::copy(a.tags.begin(), a.tags.end(), tag_index + pos);
        pos += (int)a.tags.size();
    }
    if (tag_offsets) tag_offsets[c->arcs.size()] = pos;
    return HK_OK;
}

extern "C" int hk_clip_get_tag_state(const hk_clip *c, unsigned char *alive,
                                     int *final_arc) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_get_tag_state: null clip");
    for (int i = 0; i < c->n; ++i) {
        if (alive) alive[i] = (unsigned char)c->alive[i];
        if (final_arc) final_arc[i] = c->final_arc[i];
    }
    return HK_OK;
}

extern "C" int hk_clip_build_initial_arcs(hk_clip *c) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_build_initial_arcs: null");
    std::vector<Arc> arcs;
    const bool stats = getenv("HK_STEP3_STATS") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a,
                 std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    {
        hk::ArcBuilder b(*c);
        b.collect_stats_ = stats;
        const auto t_built = now();
        // trial pass: a few confident arcs to estimate gravity
        int rc = b.construct(5, arcs);
        if (rc != HK_OK) return rc;
        hk::find_global_params(*c, arcs);
        const auto t1 = now();
        arcs.clear();
        rc = b.construct(0, arcs);
        if (rc != HK_OK) return rc;
        hk::find_global_params(*c, arcs);
        if (stats)
            fprintf(stderr,
                    "step3: set-up %.1f ms, trial pass %.1f ms, full pass %.1f ms (%ld stale "
                    "results recomputed in %.1f ms, committer ran %ld seeds, "
                    "seed time summed over threads %.1f ms, %ld seeds run, %ld "
                    "growth steps; device results used %ld, handed back %ld)\n",
                    ms(t0, t_built), ms(t_built, t1), ms(t1, now()),
                    b.stat_stale_, b.stat_stale_ms_,
                    b.stat_helped_, b.stat_busy_ms_, b.stat_runs_,
                    b.stat_steps_, b.stat_dev_used_.load(),
                    b.stat_dev_host_.load());
#ifdef HK_S3_PROF
        if (stats) {
            static const char *nm[12] = {"frontier init", "distance scan + min",
                "frontier add (growth)", "fit", "member check", "seed nb list",
                "frontier tags scanned (n)", "-", "insert + doomed",
                "frontier compaction", "whole trial", "whole seed"};
            for (int k = 0; k < 12; ++k)
                fprintf(stderr, "  s3prof %-24s %8.2f Mcyc  n=%llu\n", nm[k],
                        hk::g_s3prof.cyc[k] / 1e6, hk::g_s3prof.cnt[k]);
            static const char *en[7] = {"no tag near", "doomed", "e <= 0",
                "member too far", "too few members", "eval_arc no", "accepted"};
            for (int k = 0; k < 7; ++k) {
                fprintf(stderr, "  s3prof end %-16s trials %7llu steps %7llu  by size <4,<8,..:",
                        en[k], hk::g_s3prof.end_n[k], hk::g_s3prof.end_steps[k]);
                for (int h = 0; h < 8; ++h)
                    fprintf(stderr, " %llu", hk::g_s3prof.hist[k][h]);
                fprintf(stderr, "\n");
            }
            hk::g_s3prof = hk::S3Prof();
        }
#endif
    }
    std::stable_sort(arcs.begin(), arcs.end(), [](const Arc &p, const Arc &q) {
        return p.f_peak < q.f_peak;
    });
    for (size_t k = 0; k < arcs.size(); ++k) {
        arcs[k].id = (int)k + 1;
        arcs[k].w_off = -1;
        arcs[k].done = false;
    }
    c->arcs.swap(arcs);
    return HK_OK;
}

// EM_optimize (reference tracker.py:912-990) for a batch of clips in
// lockstep.  The reference's loop, per clip:
//     tilt, fit, merge*, prune*, E-step, fit      while anything was merged
//     clean_notes, tilt, fit, sort                or pruned
// Every device phase is one launch over the arcs of all clips that are at
// that point of their loop, and one stream synchronisation, however many
// clips that is; the order-dependent decisions (merge scan, prune, clean) stay
// per clip on the host, exactly as for a clip on its own, so the results do
// not depend on what a clip shares its batch with.  Phases whose inputs are
// already on the device follow each other without a host round trip (the
// M-step behind the E-step), and prune reads the distances the merge phase
// fetched.
// fn(j) for j = 0 .. n-1, the clips of a batch: their host phases are
// independent, so with several clips each gets a thread of its own for the
// phase (a batch is large exactly when many builds ended together, i.e. when
// the cores that built them have just come free).  Inline for one clip, and
// whenever a clip has a progress callback (which is owed the caller's
// thread).  Returns the first failure, with its message in the caller's
// error buffer.
template <class Fn>
static int for_each_clip(hk_clip *const *clips, const std::vector<int> &act,
                         Fn &&fn) {
    const size_t n = act.size();
    bool inline_only = n < 2;
    for (int ci : act) inline_only = inline_only || clips[ci]->progress;
    if (inline_only) {
        for (size_t j = 0; j < n; ++j) {
            const int rc = fn(j);
            if (rc != HK_OK) return rc;
        }
        return HK_OK;
    }
    std::vector<int> rcs(n, HK_OK);
    std::vector<std::string> msgs(n);
    auto run = [&](size_t j) {
        rcs[j] = fn(j);
        if (rcs[j] != HK_OK) msgs[j] = hk::error_buffer();
    };
    std::vector<std::thread> pool;
    pool.reserve(n - 1);
    for (size_t j = 1; j < n; ++j) pool.emplace_back(run, j);
    run(0);
    for (std::thread &t : pool) t.join();
    for (size_t j = 0; j < n; ++j)
        if (rcs[j] != HK_OK) return hk::fail(rcs[j], "%s", msgs[j].c_str());
    return HK_OK;
}

// own: the batch to work in (a clip's own batch of one, which may hold E-step
// weights from hk_clip_calculate_weights), or null to open one for the call.
static int em_optimize_batch(hk_clip *const *clips, int n, hipStream_t s,
                             hk::EmBatch *own = nullptr) {
    const bool stats = getenv("HK_STEP4_STATS") != nullptr;
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rounds = 0;
    auto timed = [&](int slot, int rc_in, auto &&fn) {
        if (rc_in != HK_OK) return rc_in;
        const auto t0 = std::chrono::steady_clock::now();
        const int r = fn();
        ms[slot] += std::chrono::duration<double, std::milli>(
                        std::chrono::steady_clock::now() - t0).count();
        return r;
    };
    hk::EmBatch *b = own;
    int rc = HK_OK;
    if (!b) rc = timed(0, HK_OK, [&] { return hk::em_open(clips, n, s, &b); });
    if (rc != HK_OK) return rc;
    std::vector<int> act((size_t)n), finishing;
    std::iota(act.begin(), act.end(), 0);
    std::vector<hk::MergeWork> work((size_t)n);
    std::vector<std::vector<double>> dist;
    std::vector<std::vector<long>> off;
    std::vector<hk::MergeInput> inputs;
    std::vector<std::vector<hk::MergeTrial>> trials;
    // act: clips at the top of their loop; finishing: clips that have left it
    // and been cleaned, with the last tilt + fit to come -- which they share
    // with the next round of the others
    while (rc == HK_OK && (!act.empty() || !finishing.empty())) {
        ++rounds;
        std::vector<int> both(act);
        both.insert(both.end(), finishing.begin(), finishing.end());
        rc = timed(1, rc, [&] { return hk::em_tilt(*b, both); });
        rc = timed(2, rc, [&] { return hk::em_fit(*b, both); });
        finishing.clear();
        if (act.empty() || rc != HK_OK) break;

        std::vector<char> again(act.size(), 0);
        for (int ci : act)
            for (Arc &a : clips[ci]->arcs) a.done = false;
        rc = timed(3, rc, [&] {
            int r = hk::em_member_distances(*b, act, dist, off);
            if (r != HK_OK) return r;
            inputs.assign(act.size(), hk::MergeInput());
            r = for_each_clip(clips, act, [&](size_t j) {
                hk::MergeWork &mw = work[(size_t)act[j]];
                hk::merge_prepare(*clips[act[j]], dist[j], off[j], mw);
                inputs[j] = mw.in;
                return (int)HK_OK;
            });
            if (r != HK_OK) return r;
            r = hk::em_merge_trials(*b, act, inputs, trials);
            if (r != HK_OK) return r;
            return for_each_clip(clips, act, [&](size_t j) {
                bool merged = false;
                const int r2 = hk::merge_scan(*clips[act[j]],
                                              work[(size_t)act[j]], trials[j],
                                              merged);
                if (merged) again[j] = 1;
                return r2;
            });
        });
        rc = timed(4, rc, [&] {
            return for_each_clip(clips, act, [&](size_t j) {
                bool pruned = false;
                const int r2 = hk::prune_phase(
                    *clips[act[j]], work[(size_t)act[j]].adist, pruned);
                if (pruned) again[j] = 1;
                return r2;
            });
        });
        rc = timed(5, rc, [&] { return hk::em_estep(*b, act, true); });
        std::vector<int> next;
        for (size_t j = 0; j < act.size(); ++j)
            (again[j] ? next : finishing).push_back(act[j]);
        rc = timed(6, rc, [&] {
            return for_each_clip(clips, finishing, [&](size_t j) {
                return hk::clean(*clips[finishing[j]]);
            });
        });
        act.swap(next);
    }
    if (!own) hk::em_close(b);
    if (rc != HK_OK) return rc;
    for (int k = 0; k < n; ++k) hk::sort_arcs_by_peak(*clips[k]);
    if (stats)
        fprintf(stderr,
                "step4: %d clips, %d rounds; open %.2f, tilt %.2f, fit %.2f, "
                "merge %.2f, prune %.2f, E+M step %.2f, clean %.2f ms\n",
                n, rounds, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6]);
    return HK_OK;
}

extern "C" int hk_clip_em_optimize(hk_clip *c, void *stream) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_em_optimize: null clip");
    // (a clip that ran single operations carries on in its own batch, where
    // its E-step weights are)
    if (c->own) c->own->stream = (hipStream_t)stream;
    return em_optimize_batch(&c, 1, (hipStream_t)stream, c->own);
}

extern "C" int hk_clips_em_optimize(hk_clip *const *clips, int nclips,
                                    void *stream) {
    if (nclips < 0 || (nclips > 0 && !clips))
        return hk::fail(HK_ERR_ARG, "hk_clips_em_optimize: bad argument");
    for (int k = 0; k < nclips; ++k) {
        if (!clips[k])
            return hk::fail(HK_ERR_ARG, "hk_clips_em_optimize: clip %d is null",
                            k);
        for (int j = 0; j < k; ++j)
            if (clips[j] == clips[k])
                return hk::fail(HK_ERR_ARG,
                                "hk_clips_em_optimize: clip %d listed twice",
                                k);
    }
    if (nclips == 0) return HK_OK;
    for (int k = 0; k < nclips; ++k)
        if (clips[k]->own)
            return hk::fail(HK_ERR_ARG,
                            "hk_clips_em_optimize: clip %d has run single EM "
                            "operations (its weights live in a batch of its "
                            "own); use hk_clip_em_optimize for it", k);
    return em_optimize_batch(clips, nclips, (hipStream_t)stream);
}

// The single operations work in the clip's own batch of one.
extern "C" int hk_clip_calculate_weights(hk_clip *c, void *stream) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_calculate_weights: null");
    hk::EmBatch *b = nullptr;
    const int rc = hk::em_own(*c, (hipStream_t)stream, &b);
    if (rc != HK_OK) return rc;
    return hk::em_estep(*b, {0}, false);
}

extern "C" int hk_clip_fit_arcs(hk_clip *c, void *stream) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_fit_arcs: null clip");
    hk::EmBatch *b = nullptr;
    const int rc = hk::em_own(*c, (hipStream_t)stream, &b);
    if (rc != HK_OK) return rc;
    return hk::em_fit(*b, {0});
}

extern "C" int hk_clip_estimate_camera_tilt(hk_clip *c, void *stream) {
    if (!c) return hk::fail(HK_ERR_ARG, "hk_clip_estimate_camera_tilt: null");
    hk::EmBatch *b = nullptr;
    const int rc = hk::em_own(*c, (hipStream_t)stream, &b);
    if (rc != HK_OK) return rc;
    return hk::em_tilt(*b, {0});
}

extern "C" int hk_clip_eval_arc(hk_clip *c, int index, int requirepeak,
                                int checkaccel) {
    if (!c || index < 0 || index >= (int)c->arcs.size())
        return hk::fail(HK_ERR_ARG, "hk_clip_eval_arc: bad arc index");
    return hk::eval_arc_host(*c, c->arcs[index], hk::rotation(*c),
                             requirepeak != 0, checkaccel != 0);
}

extern "C" int hk_clip_member_distances(hk_clip *c, double *dist,
                                        void *stream) {
    if (!c || !dist)
        return hk::fail(HK_ERR_ARG, "hk_clip_member_distances: null");
    hk::EmBatch *b = nullptr;
    int rc = hk::em_own(*c, (hipStream_t)stream, &b);
    if (rc != HK_OK) return rc;
    std::vector<std::vector<double>> d;
    std::vector<std::vector<long>> off;
    rc = hk::em_member_distances(*b, {0}, d, off);
    if (rc != HK_OK) return rc;
    std::copy(d[0].begin(), d[0].end(), dist);
    return HK_OK;
}

extern "C" int hk_release_cached_memory(void) {
    hk::seed_release_cached();
    hk::em_release_cached();
    return HK_OK;
}

// The calling thread helps a step-3 build of this process that is in its full
// pass (the one with the most seeds left per thread) for up to max_seeds
// seeds, or until that build has no seed left to hand out.  Returns 1 if it
// helped, 0 if there was nothing to help with.  Results do not depend on it
// (speculative seeds, committed in order: see ArcBuilder).
extern "C" int hk_step3_help(int max_seeds) {
    if (max_seeds <= 0) return 0;
    hk::ArcBuilder *b = nullptr;
    int slot = -1;
    {
        std::lock_guard<std::mutex> lock(hk::g_open_mu);
        double best = 64.0;             // not worth joining for a few seeds
        for (hk::ArcBuilder *cand : hk::g_open_builders) {
            const double load = cand->load_per_thread();
            if (load > best) {
                best = load;
                b = cand;
            }
        }
        if (b) slot = b->helper_join();
    }
    if (!b || slot < 0) return 0;
    b->helper_work(slot, max_seeds);
    return 1;
}

extern "C" int hk_set_host_threads(int n) {
    if (n < 0) return hk::fail(HK_ERR_ARG, "hk_set_host_threads: %d", n);
    hk::g_host_threads.store(n);
    return HK_OK;
}

extern "C" int hk_set_self_check(int on) {
    hk::g_self_check.store(on != 0);
    return HK_OK;
}
