// State of one clip between step 2 (detection) and the end of step 4 (EM):
// the SoA tag arrays, the arc list and the device mirrors used by the EM
// kernels.  Shared by hk_arcs.cpp (host control flow, steps 3-4) and
// hk_em.hip (K5/K6/K7 kernels and their launchers).
#pragma once

#include <cstdint>
#include <vector>

#include "hk_common.h"
#include "hk_fit.h"

namespace hk {

struct Arc {
    double a = 0, b = 0, c = 0, e = 0, f_peak = 0;
    int id = 0;
    bool done = false;
    std::vector<int> tags;      // member tag indices, ascending
    // E-step window: tag indices [w_lo, w_hi) hold dense weights at
    // dev.weights[w_off ...]; w_off < 0 means "unit weights on `tags`"
    int w_lo = 0, w_hi = 0;
    long w_off = -1;
};

// What the EM kernels work on (allocated on first use in step 4).
struct DeviceState {
    bool ready = false;
    int device = 0;
    int n = 0;                          // tags
    int tag_cap = 0;                    // capacity of the tag arrays
    int *frame = nullptr;
    double *x = nullptr, *y = nullptr, *r = nullptr, *tw = nullptr;
    double *total = nullptr;            // [n] per-tag weight sums
    char *h_tags = nullptr;             // pinned, tag_cap * 36 bytes: the
                                        // tag arrays on their way up
    void *uploaded = nullptr;           // hipEvent_t behind the uploads from
                                        // h_tags; waited for before the
                                        // workspace changes hands
    bool upload_pending = false;
    // per-launch arc tables (grown on demand)
    int arc_cap = 0;
    double *arc_params = nullptr;       // [arc_cap][8]
    long *arc_win = nullptr;            // [arc_cap][4]
    double *arc_out = nullptr;          // [arc_cap][8]
    double *h_arc_params = nullptr;     // pinned mirrors
    long *h_arc_win = nullptr;
    double *h_arc_out = nullptr;
    long weight_cap = 0;                // doubles
    double *weights = nullptr;          // dense E-step weights
    long bitmap_cap = 0;                // 64-bit words
    uint64_t *bitmap = nullptr;         // membership bits, one row per arc
    uint64_t *h_bitmap = nullptr;       // pinned
    long list_cap = 0;                  // ints
    int *list = nullptr;                // explicit member lists (CSR payload)
    double *list_out = nullptr;         // per list entry result (distances)
    int *h_list = nullptr;              // pinned
    double *h_list_out = nullptr;       // pinned
    size_t merge_cap = 0;               // bytes
    char *merge_buf = nullptr;          // merge-trial staging block
    char *h_merge_buf = nullptr;        // pinned
};

}  // namespace hk

struct hk_clip {
    // fixed inputs
    int W = 0, H = 0, F = 0;
    double fps = 0;
    hk_scan_params p;
    int n = 0;
    std::vector<int> frame;
    std::vector<double> x, y, r, tw;
    std::vector<int> start;             // CSR frame -> first tag index, F+1
    // evolving state
    std::vector<char> alive;            // cleared by clean()
    std::vector<int> final_arc;         // per tag: index into arcs or -1
    double tilt = 0.0;
    bool has_g = false;
    double g = 0.0, cm_per_pixel = 0.0;
    std::vector<hk::Arc> arcs;
    int device = 0;
    hk::DeviceState dev;
    // progress / abort
    hk_progress_fn progress = nullptr;
    void *progress_user = nullptr;
};

namespace hk {

// --- geometry shared by host code (hk_arcs.cpp) ----------------------------
struct Rot {
    double s, c;
};
Rot rotation(const hk_clip &clip);
void arc_position(const Arc &arc, double frame, const Rot &rot, double &px,
                  double &py);
double arc_tag_distance(const hk_clip &clip, const Arc &arc, const Rot &rot,
                        int tag);
void arc_frame_range(const hk_clip &clip, const Arc &arc, int &f_min,
                     int &f_max);
double median_tag_radius(const hk_clip &clip, const std::vector<int> &tags);

// --- device side (hk_em.hip) ------------------------------------------------
int em_prepare(hk_clip &clip, hipStream_t stream);  // upload tags once
void em_release(hk_clip &clip);      // parks the workspace for the next clip
void em_release_cached();            // frees the parked workspaces
// E-step for every arc: raw weights, per-tag totals, normalisation; fills
// arc.w_lo/w_hi/w_off and arc.tags (ascending).  (reference tracker.py:992-1039)
int em_estep(hk_clip &clip, hipStream_t stream);
// Weighted least-squares fit of every arc with >= 3 tags; updates the arcs.
// (reference tracker.py:698-798)
int em_fit(hk_clip &clip, hipStream_t stream);
// Camera tilt from the x/y accelerations of every arc; updates clip.tilt.
// (reference tracker.py:1041-1124)
int em_tilt(hk_clip &clip, hipStream_t stream);
// Distances of every arc's member tags to the arc, in arc order then member
// order (reference types.py:70-72 over `for t in arc.tags`).
int em_member_distances(hk_clip &clip, hipStream_t stream,
                        std::vector<double> &dist,
                        std::vector<long> &offsets);
// Trial fits of all candidate pairs of a merge pass (reference
// tracker.py:1199-1233), bit-identical to the host's own trial fit.
int em_merge_trials(hk_clip &clip, hipStream_t stream,
                    const std::vector<int> &good_off,
                    const std::vector<int> &good,
                    const std::vector<int> &pairs,
                    std::vector<MergeTrial> &out);
// Host decision shared by em_tilt and the arc evaluators.
bool acceleration_ok(const hk_clip &clip, double e);

}  // namespace hk
