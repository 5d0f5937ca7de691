// Device side of step 3 (hk_seed.hip) as seen by the host builder
// (hk_arcs.cpp): neighbour lists and per-seed trial arcs for a window of
// seeds at a time.  See hk_seed.hip.
#pragma once

#include <vector>

#include "hk_clip.h"

namespace hk {

constexpr int S3_NB_CAP = 128;          // neighbours per tag
constexpr int S3_MEMBERS = 256;         // tags in a trial arc
constexpr int S3_CONSULTED = 128;       // tags a seed's run may look at
constexpr int S3_MADE_CAP = 256;        // arcs found per window

enum SeedStatus {
    S3_NONE = 0,        // the seed yields no arc
    S3_MADE = 1,        // it yields the arc listed in the made_* arrays
    S3_HOST = 2,        // the fixed-size lists did not suffice: run on host
    S3_SKIPPED = 3      // the seed tag was taken already when the window ran
};

struct SeedParams {
    double e0, maxd, rot_s, rot_c, v_max;
    double accel_lo, accel_hi, accel_floor;     // acceleration_ok() bounds
    int has_g, has_vmax, gap, min_tags_to_curve_fit, min_tags_per_arc;
    int budget;         // growth steps after which a seed is left to the host
};

// Host views (pinned memory) of one window's results; valid until
// seed_close().
struct SeedRound {
    int p0 = 0, count = 0;      // seeds queue[p0 .. p0 + count)
    const int *status;          // [count]
    const int *n_cons;          // [count]
    const int *cons;            // [count][S3_CONSULTED]
    int made_count;
    const int *made_seed;       // position in the window
    const double *made_params;  // [made_count][5] a b c e f_peak
    const int *made_n;
    const int *made_members;    // [made_count][S3_MEMBERS], ascending
};

struct SeedDevice;

// *out == nullptr on return with HK_OK: the device path does not apply here
// (no GPU, or a neighbour list longer than S3_NB_CAP); use the host code.
int seed_open(hk_clip &c, const SeedParams &prm, const std::vector<int> &queue,
              const std::vector<int> &qpos, int window, SeedDevice **out);
void seed_close(SeedDevice *sd);
int seed_windows(const SeedDevice *sd);
// Window `index` (seeds queue[index * window ...)) against the snapshot
// done[0 .. n); blocks until the results are on the host.
int seed_round(SeedDevice *sd, int index, const unsigned char *done,
               SeedRound &r);
void seed_release_cached();

}  // namespace hk
