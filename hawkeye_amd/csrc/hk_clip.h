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

// One row per clip of a launch, read by the EM kernels: step 4 of several
// clips runs in lockstep, every device phase as ONE launch over the arcs of
// all of them (BASELINE configs[4]: EM batched across clips).
struct ClipDev {
    double sn, cs;              // sin / cos of the clip's camera tilt
    double sigmasq, maxd;       // E-step sigma^2, max_distance_pixels
    int tag_base, n_tags;       // the clip's tags in the batch's tag arrays
    int arc_base, n_arcs;       // its rows in this launch's arc tables
    int min_tags, pad;
};

// What the EM kernels work on: the device workspace of a batch of clips
// (allocated on first use in step 4, pooled per device).
struct DeviceState {
    int device = 0;
    int n = 0;                          // tags of all clips of the batch
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
    void *waited = nullptr;             // hipEvent_t of the blocking stream
                                        // waits (wait_for_stream)
    // per-launch arc tables (grown on demand)
    int arc_cap = 0;
    double *arc_params = nullptr;       // [arc_cap][8]
    long *arc_win = nullptr;            // [arc_cap][4]
    double *arc_out = nullptr;          // [arc_cap][8]
    int *arc_clip = nullptr;            // [arc_cap] row of clip_tab
    double *h_arc_params = nullptr;     // pinned mirrors
    long *h_arc_win = nullptr;
    double *h_arc_out = nullptr;
    int *h_arc_clip = nullptr;
    int clip_cap = 0;
    ClipDev *clip_tab = nullptr;        // [clip_cap] the clips of the launch
    ClipDev *h_clip_tab = nullptr;      // pinned
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

struct EmBatch;

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
    // the batch of one that the single-operation entry points
    // (hk_clip_calculate_weights ...) work in; hk_clip_em_optimize and
    // hk_clips_em_optimize open a batch of their own
    hk::EmBatch *own = nullptr;
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
// A batch: clips of one device whose step 4 runs in lockstep.  Every phase
// below takes `act`, the clips (positions in the batch) it is to run for, and
// is one launch (or one chain of launches) and one stream synchronisation
// however many clips that is.  Arc::w_off refers to the batch's weight buffer.
struct EmBatch {
    std::vector<hk_clip *> clips;
    std::vector<int> tag_base;          // per clip: its first tag in the batch
    int device = 0;
    hipStream_t stream = nullptr;
    DeviceState dev;
    // the arc tables staged last: first row of each clip of `act`
    std::vector<int> arc_base;
    int rows = 0;
};
// Takes a workspace over and uploads the clips' tags (asynchronously).
int em_open(hk_clip *const *clips, int n, hipStream_t stream, EmBatch **out);
void em_close(EmBatch *batch);      // parks the workspace for the next batch
void em_release(hk_clip &clip);     // closes the clip's own batch, if any
void em_release_cached();           // frees the parked workspaces
// The clip's own batch of one (opened on first use).
int em_own(hk_clip &clip, hipStream_t stream, EmBatch **out);
// E-step for every arc: raw weights, per-tag totals, normalisation; fills
// arc.w_lo/w_hi/w_off and arc.tags (ascending).  (reference
// tracker.py:992-1039)  With fit_too the M-step runs behind it in the same
// chain of launches (the weights never leave the device in between).
int em_estep(EmBatch &b, const std::vector<int> &act, bool fit_too);
// Weighted least-squares fit of every arc with >= 3 tags; updates the arcs.
// (reference tracker.py:698-798)
int em_fit(EmBatch &b, const std::vector<int> &act);
// Camera tilt from the x/y accelerations of every arc; updates clip.tilt.
// (reference tracker.py:1041-1124)
int em_tilt(EmBatch &b, const std::vector<int> &act);
// Distances of every arc's member tags to the arc, per clip of `act` in arc
// order then member order (reference types.py:70-72 over `for t in arc.tags`).
int em_member_distances(EmBatch &b, const std::vector<int> &act,
                        std::vector<std::vector<double>> &dist,
                        std::vector<std::vector<long>> &offsets);
// Trial fits of all candidate pairs of a merge pass (reference
// tracker.py:1199-1233), bit-identical to the host's own trial fit.  Per clip
// of `act`: the arcs' good-tag lists (CSR over the clip's arc list, local tag
// indices) and the pairs (positions in that list); clips without pairs take
// no part.
struct MergeInput {
    std::vector<int> good_off, good, pairs;
};
int em_merge_trials(EmBatch &b, const std::vector<int> &act,
                    const std::vector<MergeInput> &in,
                    std::vector<std::vector<MergeTrial>> &out);
// Host decision shared by em_tilt and the arc evaluators.
bool acceleration_ok(const hk_clip &clip, double e);

}  // namespace hk
