/*
 * hk.h -- C ABI of libhawkeye_hip.so, the MI355X (gfx950) implementation of
 * Hawkeye's detection + arc-fitting hot path.
 *
 * The reference (jkboyce/hawkeye) has no FFI of its own: the boundary of the
 * hot path is the Python class API of hawkeye/tracker.py.  Each entry point
 * below therefore cites the reference call site whose arithmetic it replaces;
 * hawkeye_amd/tracker.py binds them with ctypes and keeps the reference's
 * Python surface on top.
 *
 * Conventions
 *   - every function returns 0 on success and a negative hk_status on error;
 *     hk_last_error() then returns a thread-local description.  No C++
 *     exception crosses this boundary.
 *   - "dev" pointers are device (HBM) addresses on the device the handle was
 *     created for; "host" pointers are ordinary host memory.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 *     Functions that only enqueue work say so; the others synchronise the
 *     stream before returning host results.
 *   - multi-GPU: one process per GPU, every process with its own handles.
 *     Nothing in this library talks to another GPU.  The exchange step of the
 *     time-segmented layout (per-rank event lists -> rank 0, 16-byte records
 *     over RCCL) and the sharding of clips over ranks live above the C ABI,
 *     in hawkeye_amd/dist.py (torch.distributed); a host in another language
 *     does that one gather with its own RCCL binding.
 *   - handles are not thread safe; use one host thread per handle (the
 *     reference drives the scanner from a single worker thread,
 *     reference worker.py:718-778).
 */
#ifndef HK_H
#define HK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum hk_status {
    HK_OK = 0,
    HK_ERR_ARG = -1,        /* bad argument / unsupported shape */
    HK_ERR_HIP = -2,        /* a HIP runtime call failed */
    HK_ERR_CAPACITY = -3,   /* an output or work list overflowed its capacity */
    HK_ERR_ALLOC = -4,
    HK_ERR_ABORTED = -5,    /* progress callback asked to stop */
    HK_ERR_INTERNAL = -6
};

const char *hk_last_error(void);
int hk_version(void);

/* Number of HIP devices visible, or a negative status. */
int hk_device_count(void);

/* A finished EM batch (hk_clip_em_optimize, hk_clips_em_optimize) parks its
 * device workspace (a few MB) for the next batch on the same device instead
 * of freeing it; this frees what is parked. */
int hk_release_cached_memory(void);

/* ------------------------------------------------------------------------
 * K1  MOG background subtraction
 *     replaces cv2.bgsegm.createBackgroundSubtractorMOG()  (reference
 *     tracker.py:357) and fgbg.apply(frame)                 (tracker.py:395)
 *     with OpenCV's default parameters: history 200, 5 mixtures,
 *     backgroundRatio 0.7, noiseSigma 15, learning rate 1/min(n, 200).
 * ---------------------------------------------------------------------- */
typedef struct hk_mog hk_mog;

/* Allocates the zero-initialised mixture model for width x height BGR frames
 * (40 fp32 planes of width*height, SoA) on `device`. */
int hk_mog_create(int width, int height, int device, hk_mog **out);
int hk_mog_destroy(hk_mog *m);
/* Back to the freshly created state (model zero, frame counter 0).  Enqueues. */
int hk_mog_reset(hk_mog *m, void *stream);
/* Frames applied so far. */
int hk_mog_nframes(const hk_mog *m);

/* One frame, = fgmask = fgbg.apply(frame).  bgr_dev: uint8 [H][bgr_pitch]
 * with 3 bytes per pixel; mask_dev: uint8 [H][mask_pitch], written with
 * {0, 255}.  Enqueues on `stream`. */
int hk_mog_apply(hk_mog *m, const void *bgr_dev, long bgr_pitch,
                 void *mask_dev, long mask_pitch, void *stream);

/* `nframes` consecutive frames in one launch; results are identical to
 * nframes calls of hk_mog_apply.  The model of each pixel stays in registers
 * for the whole batch, so HBM traffic is ~4 B/pixel/frame instead of 324.
 * Frame t is at bgr_dev + t*bgr_frame_stride (bytes), mask t at
 * mask_dev + t*mask_frame_stride.  Enqueues on `stream`. */
int hk_mog_apply_batch(hk_mog *m, const void *bgr_dev, long bgr_frame_stride,
                       long bgr_pitch, void *mask_dev, long mask_frame_stride,
                       long mask_pitch, int nframes, void *stream);

/* The same launch with the foreground mask written as 1 bit per pixel for the
 * blob detector (hk_blob_tiles_buffer / hk_blob_enqueue_tiles) instead of a
 * byte image: per frame and 32x4 pixel tile (tiles in raster order) four
 * 32-bit words, the bits of the tile's even pixels (16 per row, rows 0-3) and
 * then of its odd pixels.  Same model update, same mask bits as
 * hk_mog_apply_batch.  Needs width % 32 == 0, height % 4 == 0
 * (hk_mog_tiles_supported) and densely packed, 2-byte aligned frames
 * (bgr_pitch == 3*width) less than 512 MB apart (bgr_frame_stride);
 * tiles_dev 16-byte aligned, frame t at tiles_dev + t * (W/32)*(H/4)*4 words.
 * No counterpart in the reference (its fgmask is a byte image,
 * tracker.py:395). */
int hk_mog_tiles_supported(const hk_mog *m);
int hk_mog_apply_batch_tiles(hk_mog *m, const void *bgr_dev,
                             long bgr_frame_stride, long bgr_pitch,
                             void *tiles_dev, int nframes, void *stream);

/* hk_mog_apply_batch for callers that cannot allocate device memory.  It has
 * NO counterpart in the reference; it exists for callers that cannot call
 * hipMalloc at all: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  It calls no hipMalloc / hipFree or any
 * other allocating routine.
 *
 * The caller owns the model: model_dev = float [40][width*height] on `device`
 * (plane = field*5 + k, fields weight, sortKey, mean[3], var[3]; all zero for
 * a fresh model), updated in place; frames_before = frames that model has
 * seen (0 for a fresh one).  The other arguments are hk_mog_apply_batch's.
 * What that entry point keeps in buffers of its handle besides the model --
 * the learning-rate table and the per-group cost / launch order lists of the
 * batch kernel -- lives here in a static device array of the library
 * (128 KB; each GPU the library runs on holds it until the process ends),
 * which limits frames to 3840x2160 pixels: larger ones are rejected with
 * HK_ERR_CAPACITY.  Results are identical to hk_mog_apply_batch (the launch
 * order is scheduling only).
 * Every call on a GPU shares that GPU's array, which keeps the previous
 * call's contents: the call re-initialises what it reads, but calls that
 * OVERLAP on one GPU (two host threads, or two streams with both launches in
 * flight) race on it, and that race can CORRUPT RESULTS, not just the launch
 * order: a batch kernel takes its pixel groups from the order list, and if a
 * call on another stream rewrites that list for its own geometry between
 * this call's ordering pass and its batch kernel, groups are skipped or
 * stepped twice.  (Within one call the list is always a permutation, whatever
 * happens to the costs it is made from; the learning-rate table is rewritten
 * with the same values by every call.)  Serialise the calls on a device, e.g.
 * by keeping them on one stream.
 * The first call on a device looks the array's device address up
 * (hipGetSymbolAddress); make that call outside stream capture.
 * Enqueues on `stream`. */
int hk_mog_apply_batch_static(int width, int height, int device,
                              void *model_dev, int frames_before,
                              const void *bgr_dev, long bgr_frame_stride,
                              long bgr_pitch, void *mask_dev,
                              long mask_frame_stride, long mask_pitch,
                              int nframes, void *stream);

/* Copies the model to host in OpenCV's AoS order: float [H*W][5][8] =
 * (sortKey, weight, mean[3], var[3]).  For tests.  Synchronises. */
int hk_mog_export_model(hk_mog *m, float *host_out, void *stream);

/* The packed fast path of the batch kernel replaces the IEEE square root and
 * divisions of process8uC3 by the same correctly rounded sequences without
 * their range scaling (operand ranges are known there).  This compares them
 * with the IEEE operators exhaustively over those ranges on `device` and
 * returns the number of differing results (must be 0).  For tests.
 * Synchronises. */
int hk_mog_selfcheck_math(int device, long long *mismatches_out);

/* ------------------------------------------------------------------------
 * K2-K4  blob detection on binary masks
 *     replaces cv2.SimpleBlobDetector_create(params).detect(fgmask)
 *     (reference tracker.py:359-374, 396): findContours(RETR_LIST,
 *     CHAIN_APPROX_NONE) + contour moments / arcLength / colour test / median
 *     radius per contour, then the cross-threshold grouping loop, returning
 *     float32 keypoints (x, y, size) in OpenCV's order.
 *
 *     Supported parameter space = what Hawkeye uses: filterByArea and
 *     filterByCircularity on (they bound the contour length, which bounds the
 *     device trace), filterByInertia and filterByConvexity off.  The input
 *     must be a binary mask (every byte 0 or 255, as MOG writes it); then all
 *     binarisation thresholds see the same image.  Anything else is rejected
 *     with HK_ERR_ARG.
 * ---------------------------------------------------------------------- */
typedef struct hk_blob_params {
    float thresholdStep, minThreshold, maxThreshold;
    int minRepeatability;
    float minDistBetweenBlobs;
    int filterByColor;
    int blobColor;
    int filterByArea;
    float minArea, maxArea;
    int filterByCircularity;
    float minCircularity, maxCircularity;
    int filterByInertia;
    int filterByConvexity;
} hk_blob_params;

/* The block reference tracker.py:359-374 builds for a scan video of the given
 * height (OpenCV defaults + Hawkeye's overrides). */
int hk_blob_params_hawkeye(hk_blob_params *p, int scan_height);

typedef struct hk_blob hk_blob;

/* Work buffers for batches of up to max_frames masks of width x height. */
int hk_blob_create(int width, int height, int max_frames,
                   const hk_blob_params *params, int device, hk_blob **out);
int hk_blob_destroy(hk_blob *b);

/* Enqueue detection on `nframes` masks (mask t at mask_dev + t*frame_stride,
 * rows `pitch` bytes apart) on `stream`.  Returns immediately.  The masks
 * must stay valid and unchanged until hk_blob_finish() returns: a batch whose
 * work lists overflow (very noisy masks) is run again there with larger
 * lists, as OpenCV has no limit on the number of contours. */
int hk_blob_enqueue(hk_blob *b, const void *mask_dev, long frame_stride,
                    long pitch, int nframes, void *stream);
/* The same, fed by the MOG kernel's 1-bit output instead of byte masks (1/8 of
 * the bytes; no counterpart in the reference, whose fgmask is a byte image:
 * tracker.py:395-396).  hk_blob_tiles_buffer() returns the handle's device
 * buffer, which hk_mog_apply_batch_tiles() fills for up to max_frames frames;
 * hk_blob_enqueue_tiles() then runs detection on its first nframes frames.
 * Frames must be made of 32x4 tiles (width % 32 == 0, height % 4 == 0). */
int hk_blob_tiles_buffer(hk_blob *b, void **tiles_dev, long *words_per_frame);
int hk_blob_enqueue_tiles(hk_blob *b, int nframes, void *stream);
/* Wait for the enqueued batch and finish on the host (ordering, radius,
 * grouping).  counts_host[t] = keypoints of frame t; kp_host receives
 * (x, y, size) float triples, frame after frame; *total = sum of counts.
 * HK_ERR_CAPACITY if total > cap: nothing beyond cap is written, *total holds
 * the size needed, and the batch's results are kept until the next
 * hk_blob_finish*() call (with a larger buffer) or the next enqueue. */
int hk_blob_finish(hk_blob *b, int *counts_host, float *kp_host, int cap,
                   int *total);
/* hk_blob_enqueue + hk_blob_finish. */
int hk_blob_detect(hk_blob *b, const void *mask_dev, long frame_stride,
                   long pitch, int nframes, void *stream, int *counts_host,
                   float *kp_host, int cap, int *total);

/* Per-contour stage only (one findBlobs() pass), for tests: centres that
 * survive the filters as double (x, y, radius) triples, per frame in OpenCV's
 * contour order.  Same calling convention as hk_blob_finish. */
int hk_blob_finish_centers(hk_blob *b, int *counts_host, double *centers_host,
                           int cap, int *total);

/* ------------------------------------------------------------------------
 * H1  synthetic frames (input generation for tests and benchmarks; the
 *     reference ships no clips, reference README.md:25-27).  Renders frames
 *     first_frame .. first_frame+nframes-1 of a scene into bgr_dev: gradient +
 *     fixed texture background, optional brightness drift, optional dark
 *     "juggler" rectangle, the discs listed per frame in discs_dev
 *     (float [total_frames][max_discs][4] = x, y, radius, kind; radius <= 0
 *     = unused; indexed by absolute frame number) and Gaussian sensor noise.
 *     Deterministic in (seed, frame, pixel).  Enqueues on `stream`.
 * ---------------------------------------------------------------------- */
int hk_render_frames(int width, int height, float fps, unsigned seed,
                     float noise_sigma, float drift_amp, float drift_hz,
                     int juggler, const float *discs_dev, int max_discs,
                     int first_frame, int nframes, void *bgr_dev,
                     long frame_stride, long pitch, void *stream);

/* ------------------------------------------------------------------------
 * Steps 3-4: preliminary arcs and EM refinement on the detected events
 *     replaces VideoScanner.build_initial_arcs()   (reference tracker.py:450-696,
 *     880-906) and VideoScanner.EM_optimize()      (reference tracker.py:912-1435).
 *
 * A "clip" holds the events of one video as SoA arrays in (frame, position
 * in notes['meas'][frame]) order.  Step 3 is host code.  Step 4 keeps the
 * reference's merge / prune / clean control flow on the host and runs
 *   K5 E-step      calculate_weights   (reference tracker.py:992-1039)
 *   K6 M-step      fit_arcs            (reference tracker.py:698-798)
 *      tilt        estimate_camera_tilt (reference tracker.py:1041-1124)
 *   K7 distances   Ballarc.get_distance_from_tag over arc members
 *                                      (reference types.py:70-72)
 *   K8 merge trials  the trial fits of all candidate arc pairs of a
 *                  merge_arcs pass    (reference tracker.py:1199-1233)
 * as fp64 HIP kernels on `device`, one wavefront per arc.
 * ---------------------------------------------------------------------- */
typedef struct hk_scan_params {
    /* reference tracker.py:2510-2570 (constants) */
    double sigmasq;
    int max_frame_gap_in_arc;
    double g_window;
    int min_tags_to_curve_fit;
    double default_frame_height_cm;
    double max_frame_height_cm;
    double default_fps;
    double max_fps;
    /* reference tracker.py:345-355 (derived in step 2) */
    int min_tags_per_arc;
    double max_distance_pixels;
    double radius_window;
} hk_scan_params;

/* Defaults of reference default_scanner_params() with the step-2 derived keys
 * for a source video of frame_height at fps scanned at scan_height. */
int hk_scan_params_default(hk_scan_params *p, int frame_height,
                           int scan_height, double fps);

/* Called from inside long loops on the calling thread; a non-zero return
 * aborts the step with HK_ERR_ABORTED (mirrors callback() in reference
 * tracker.py:630-631, 1283-1284, 1325-1326, 1375-1376). */
typedef int (*hk_progress_fn)(void *user);

typedef struct hk_clip hk_clip;

/* Copies the events.  frame[] must be ascending in [0, frame_count).
 * No HIP call is made until a step-4 function runs. */
int hk_clip_create(int frame_width, int frame_height, double fps,
                   int frame_count, const hk_scan_params *params, int ntags,
                   const int *frame, const double *x, const double *y,
                   const double *radius, const double *total_weight,
                   int device, hk_clip **out);
int hk_clip_destroy(hk_clip *c);
int hk_clip_set_progress(hk_clip *c, hk_progress_fn fn, void *user);

/* notes['camera_tilt'] and notes['g_px_per_frame_sq'] (has_g = key present). */
int hk_clip_set_state(hk_clip *c, double camera_tilt, int has_g, double g);
int hk_clip_get_state(const hk_clip *c, double *camera_tilt, int *has_g,
                      double *g, double *cm_per_pixel);

/* notes['arcs']: params5 = (a, b, c, e, f_peak) per arc; members of arc i are
 * tag_index[tag_offsets[i] .. tag_offsets[i+1]).  Weights are 1.0, as after
 * step 3. */
int hk_clip_set_arcs(hk_clip *c, int narcs, const double *params5,
                     const int *ids, const int *tag_offsets,
                     const int *tag_index);
int hk_clip_arc_count(const hk_clip *c);
int hk_clip_arc_tag_total(const hk_clip *c);
int hk_clip_get_arcs(const hk_clip *c, double *params5, int *ids,
                     int *tag_offsets, int *tag_index);
/* alive[i] = 0 for events removed by clean_notes; final_arc[i] = index of
 * tag.arc in the arc list, or -1. */
int hk_clip_get_tag_state(const hk_clip *c, unsigned char *alive,
                          int *final_arc);

/* Step 3 (host only; multi-threaded, same result for any thread count). */
int hk_clip_build_initial_arcs(hk_clip *c);
/* Batches: the calling thread lends a hand to a step-3 build that another
 * thread of this process is running (hk_clip_build_initial_arcs in its full
 * pass), for up to max_seeds seeds.  Returns 1 if it helped, 0 if no build
 * had work to share.  The build's result does not depend on it.  No
 * counterpart in the reference (one thread, tracker.py:487-652). */
int hk_step3_help(int max_seeds);

/* 1 if this build of the library contains the opt-in device side of step 3
 * (hk_seed.hip; `make DEVICE_SEEDS=1` builds it as a second library,
 * libhawkeye_hip_device_seeds.so; selected at run time with
 * HK_STEP3_DEVICE=1|2), 0 for the product library, in which step 3 is host
 * code only.  Either way the arcs are the same.  No counterpart in the reference. */
int hk_step3_device_built(void);

/* Host threads step 3 may use: 0 = automatic (the smaller of the hardware
 * concurrency, OMP_NUM_THREADS and 16), else exactly n (at most 16). */
int hk_set_host_threads(int n);
/* Step 4.  Synchronises `stream` as needed. */
int hk_clip_em_optimize(hk_clip *c, void *stream);
/* Step 4 of several clips (of one device) in lockstep: EM_optimize
 * (tracker.py:912-990) runs once per clip in the reference; here every device
 * phase -- tilt, M-step, member distances, merge trial fits, E-step -- is ONE
 * launch over the arcs of all clips that are at that point of their loop
 * (BASELINE configs[4]: "EM batched across clips"), while the order-dependent
 * merge / prune / clean decisions stay per clip.  Each clip ends up exactly
 * as hk_clip_em_optimize leaves it on its own.  hk_clip_em_optimize is the
 * batch of one.  Clips that have run the single operations below are refused
 * (HK_ERR_ARG). */
int hk_clips_em_optimize(hk_clip *const *clips, int nclips, void *stream);

/* Debug aid: with on != 0 (or HK_SELF_CHECK set in the environment) every
 * merge trial fit that step 4 takes from the device (K8) is recomputed on the
 * host and must match bit for bit, else the step fails with
 * HK_ERR_INTERNAL. */
int hk_set_self_check(int on);

/* The individual EM operations, as VideoScanner exposes them. */
int hk_clip_calculate_weights(hk_clip *c, void *stream);
int hk_clip_fit_arcs(hk_clip *c, void *stream);
int hk_clip_estimate_camera_tilt(hk_clip *c, void *stream);
/* Normalised E-step weights as (tag, arc index, weight) triples in arc order,
 * after hk_clip_calculate_weights.  Returns the count (may exceed cap). */
int hk_clip_get_weights(hk_clip *c, int *tag, int *arc, double *weight,
                        int cap, void *stream);
/* eval_arc() failure code of arc `index` (reference tracker.py:800-840). */
int hk_clip_eval_arc(hk_clip *c, int index, int requirepeak, int checkaccel);
/* Distances of every arc's members to the arc (K7), laid out like
 * hk_clip_get_arcs' tag_index. */
int hk_clip_member_distances(hk_clip *c, double *dist, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HK_H */
