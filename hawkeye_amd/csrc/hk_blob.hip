// K2-K4 -- SimpleBlobDetector on binary masks for gfx950.
//
// Replaces detector.detect(fgmask) (reference tracker.py:396; arithmetic in
// OpenCV's blobdetector.cpp / contours.cpp / moments.cpp / shapedescr.cpp,
// restated in oracle/detect_ref.c).
//
// OpenCV finds borders with a sequential raster scan that marks pixels as it
// traces, so that every border is followed exactly once.  Here the same set
// of borders is found without marks:
//
//   K2 blob_scan_k   reads each mask byte once (16 B per lane, HBM-bound) and
//                    emits *candidate* border starts: a foreground pixel
//                    whose W, NW, N, NE neighbours are background (possible
//                    raster-first pixel of an 8-connected component), or a
//                    background pixel whose W and N neighbours are
//                    foreground (possible raster-first pixel of a 4-connected
//                    hole).  Wave-level ballot/prefix compaction, one atomic
//                    per wave.
//   K3 blob_trace_k  one workgroup per (frame, band of rows) copies the
//                    band's 1-bit image (written by K2) into LDS; one lane per
//                    candidate then follows its border exactly as
//                    icvFetchContour does (same start direction, same sweep
//                    order), accumulating the integer Green sums and
//                    straight/diagonal step counts.  A candidate is the real
//                    start iff nothing on its border lies before it in raster
//                    order; other candidates retire.  The area and
//                    circularity filters bound the useful contour length
//                    (perimeter^2 <= 4 pi maxArea / minCircularity), so the
//                    trace is cut there.  Survivors get their centroid
//                    (double), pass the colour test and store their points.
//   K3b blob_radius_k one wave per surviving contour selects the two middle
//                    order statistics of the squared centre-to-point
//                    distances (rank counting in LDS).
//   K4 (host)        orders contours as findContours(RETR_LIST) returns them,
//                    takes the square roots, replays the cross-threshold
//                    grouping loop of SimpleBlobDetector::detect and emits
//                    float32 keypoints.
//
// All integer work is exact; the few double operations are IEEE and
// uncontracted, so results equal the oracle's bit for bit.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "hk_common.h"

namespace {

constexpr int MAXPTS = 256;             // hard cap on a traced contour
constexpr int CTR_NACTIVE = 0, CTR_NBLOB = 1, CTR_FLAGS = 2, CTR_NTMP = 3,
              CTR_NEXT = 4, CTR_WORDS = 8;
constexpr unsigned FLAG_NONBINARY = 1u, FLAG_BAND_FULL = 2u;
constexpr unsigned BAND_CAP0 = 512;     // candidates per (frame, band, block) to
                                        // start with; grown on demand
constexpr int TRACE_LDS_BYTES = 128 * 1024;
// lanes of a border-tracing workgroup: a band holds a dozen candidates on
// average, and every wave of the workgroup holds a wave slot (and the band's
// LDS) for as long as the band's longest border takes
constexpr int TRACE_THREADS = 64;
// A tracing workgroup works on the candidates of a block of COL_WORDS words
// of a band and holds a window of WIN_WORDS words per row in LDS: the block
// plus COL_MARGIN words (128 pixels) on either side.  A closed border of at
// most max_pts <= 256 points stays within max_pts / 2 pixels of its start in
// every direction, and a trace that goes further is given up.
// The window starts at a multiple of four words (rows of the 1-bit image are
// padded to a multiple of four words, so that the window is filled with
// 16-byte loads): 5 words of margin on the left, 7 on the right.
constexpr int COL_WORDS = 8, COL_MARGIN = 4;
constexpr int WIN_WORDS = 20;
static_assert(WIN_WORDS % 4 == 0 && WIN_WORDS >= COL_WORDS + 2 * COL_MARGIN + 2,
              "window: block + margins, whole 16-byte groups");

struct Cand {
    uint32_t pos;           // (y << 16) | x of the start (foreground) pixel
    uint32_t fh;            // (frame << 1) | is_hole
};

struct BlobRec {
    int32_t frame;
    uint32_t key;           // (y << 16) | x where the raster scan finds it
    int32_t is_hole;
    int32_t npts;
    int32_t pts_at;         // >= 0: slot of the trace's point list in the
                            // first-pass buffer; -1: pts[own slot]
    int32_t pad_;
    long long a00, a10, a01;
    double cx, cy;
    double d2lo, d2hi;      // middle order statistics of squared distances
};

// What the host reads back per surviving contour (written by blob_radius_k).
struct HostRec {
    uint32_t key;           // as BlobRec::key
    int32_t frame;
    double cx, cy, radius;  // radius = mean of the two middle distances
};

struct DevParams {
    int filterByColor, blobColor;
    double minArea, maxArea, minCirc, maxCirc;
    int max_pts;
};

__device__ __forceinline__ int dir_dx(int s) {
    return (int)((0x901Au >> (2 * s)) & 3u) - 1;    // E NE N NW W SW S SE
}
__device__ __forceinline__ int dir_dy(int s) {
    return (int)((0xA901u >> (2 * s)) & 3u) - 1;
}

// top bits of 4 bytes -> 4 bits, byte 0 -> bit 0
__device__ __forceinline__ unsigned top_bits(unsigned w) {
    return ((((w >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ bool not_binary(unsigned w) {
    const unsigned hi = (w >> 7) & 0x01010101u;
    return (w & 0x7F7F7F7Fu) != hi * 0x7Fu;
}

// Foreground bits of pixels x0 .. x0+15 of one row (bit i = pixel x0+i);
// pixels outside [0, W) read as background.
__device__ __forceinline__ unsigned row_bits16(const uint8_t *row, int x0,
                                               int W, bool vec,
                                               bool &nonbinary) {
    if (row == nullptr) return 0u;
    if (vec && x0 + 16 <= W) {
        const uint4 v = *reinterpret_cast<const uint4 *>(row + x0);
        nonbinary |= not_binary(v.x) || not_binary(v.y) || not_binary(v.z) ||
                     not_binary(v.w);
        return top_bits(v.x) | (top_bits(v.y) << 4) | (top_bits(v.z) << 8) |
               (top_bits(v.w) << 12);
    }
    unsigned bits = 0;
    for (int i = 0; i < 16; ++i) {
        const int x = x0 + i;
        if (x < W) {
            const unsigned b = row[x];
            nonbinary |= (b != 0u && b != 255u);
            bits |= (b >> 7) << i;
        }
    }
    return bits;
}
__device__ __forceinline__ unsigned row_bit(const uint8_t *row, int x, int W) {
    return (row != nullptr && x >= 0 && x < W) ? (unsigned)(row[x] >> 7) : 0u;
}

// Candidates that cannot be border starts, found without tracing.  OpenCV
// starts a border at the first transition of that border in raster order
// (every pixel of a followed border is marked, contours.cpp), so:
//  * an outer candidate (first pixel of a run of foreground pixels in its
//    row, nothing above its own three neighbours) is not a start if ANY pixel
//    of the run touches a foreground pixel of the row above: the component
//    has an earlier pixel, and whichever border the candidate lies on was
//    followed from there;
//  * a hole candidate (first pixel of a run of background pixels, foreground
//    to its left and above it) is not a start if any pixel of the run has a
//    background pixel above it: that pixel belongs to the same background
//    region and the border had a transition in the row above.  (This is what
//    the lower-right side of every blob produces: ~4 such candidates per blob,
//    each of which the tracer used to follow half-way round the blob.)
// `row` / `up`: bit j = pixel (candidate column - i + j) of the candidate's
// row / the row above, `avail` bits of them known (pixels beyond the image
// count as background and may be included); a run that leaves the known bits
// undecided keeps its candidate.  Removes only candidates whose trace would
// have been discarded, so results do not change.
__device__ __forceinline__ void prune_candidates(unsigned &outer,
                                                 unsigned &hole,
                                                 unsigned long long row,
                                                 unsigned long long up,
                                                 int avail) {
    unsigned o = outer, h = hole;
    while (o) {
        const int i = __ffs(o) - 1;
        o &= o - 1;
        const unsigned long long v = ~row >> i;
        int len = v ? __ffsll((long long)v) - 1 : 64;
        const int left = avail - i;
        if (len > left) len = left;
        const int span = len + 1 > left ? left : len + 1;
        const unsigned long long m =
            span >= 64 ? ~0ull : (1ull << span) - 1ull;
        if ((up >> i) & m) outer &= ~(1u << i);
    }
    while (h) {
        const int i = __ffs(h) - 1;
        h &= h - 1;
        const unsigned long long v = row >> i;
        int len = v ? __ffsll((long long)v) - 1 : 64;
        const int left = avail - i;
        if (len > left) len = left;
        const unsigned long long m = len >= 64 ? ~0ull : (1ull << len) - 1ull;
        if (~(up >> i) & m) hole &= ~(1u << i);
    }
}

// ---------------------------------------------------------------------------
// K2
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blob_scan_k(
    const uint8_t *__restrict__ masks, long frame_stride, long pitch, int W,
    int H, int nframes, int segs, bool vec, Cand *__restrict__ cands,
    unsigned *__restrict__ ctr, int band_rows, int nbands,
    unsigned short *__restrict__ bits, int words_per_row, unsigned band_cap,
    int ncols, unsigned *__restrict__ active) {
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)nframes * H * segs;
    if (tid >= total) return;
    bool nonbinary = false;
    const int seg = (int)(tid % segs);
    const long rowid = tid / segs;
    const int y = (int)(rowid % H);
    const int f = (int)(rowid / H);
    const int x0 = seg * 16;
    const uint8_t *row = masks + (long)f * frame_stride + (long)y * pitch;
    const uint8_t *prow = y > 0 ? row - pitch : nullptr;
    // bit j of cur17 / up18 = pixel x0 - 1 + j
    const unsigned cur17 = (row_bits16(row, x0, W, vec, nonbinary) << 1) |
                           row_bit(row, x0 - 1, W);
    bool dummy = false;
    const unsigned up18 = (row_bits16(prow, x0, W, vec, dummy) << 1) |
                          row_bit(prow, x0 - 1, W) |
                          (row_bit(prow, x0 + 16, W) << 17);
    const int nvalid = W - x0 < 16 ? W - x0 : 16;
    const unsigned valid = (1u << nvalid) - 1u;
    // the 1-bit image the border tracer reads: padded by one zero row
    // above/below and one zero 32-bit word left/right (never written)
    bits[((long)f * (H + 2) + y + 1) * words_per_row * 2 + 2 + seg] =
        (unsigned short)((cur17 >> 1) & valid);
    unsigned outer = (cur17 >> 1) & ~cur17 & ~up18 & ~(up18 >> 1) &
                     ~(up18 >> 2) & valid;
    unsigned hole = ~(cur17 >> 1) & cur17 & (up18 >> 1) & valid;
    if (nonbinary) atomicOr(&ctr[CTR_FLAGS], FLAG_NONBINARY);
    // (this lane knows its 16 pixels and pixel x0 + 16 of the row above)
    if (outer | hole)
        prune_candidates(outer, hole, cur17 >> 1, up18 >> 1,
                         x0 + 16 >= W ? 64 : 16);

    // candidates go to the list of their (frame, band of rows, block of
    // columns); few 16-pixel segments hold any, so one atomic per such
    // segment is cheap
    const unsigned c = __popc(outer) + __popc(hole);
    if (c == 0) return;
    const unsigned bucket =
        ((unsigned)f * nbands + (unsigned)(y / band_rows)) * ncols +
        (unsigned)(x0 / (32 * COL_WORDS));
    unsigned at = atomicAdd(&ctr[CTR_WORDS + bucket], c);
    if (at == 0u)       // the bucket's first candidates: list it for tracing
        active[atomicAdd(&ctr[CTR_NACTIVE], 1u)] = bucket;
    if (at + c > band_cap) atomicOr(&ctr[CTR_FLAGS], FLAG_BAND_FULL);
    Cand *list = cands + (size_t)bucket * band_cap;
    while (outer) {
        const int i = __ffs(outer) - 1;
        outer &= outer - 1;
        if (at < band_cap)
            list[at] = Cand{((unsigned)y << 16) | (unsigned)(x0 + i),
                            (unsigned)f << 1};
        ++at;
    }
    while (hole) {
        const int i = __ffs(hole) - 1;
        hole &= hole - 1;
        if (at < band_cap)      // origin = the foreground pixel left of it
            list[at] = Cand{((unsigned)y << 16) | (unsigned)(x0 + i - 1),
                            ((unsigned)f << 1) | 1u};
        ++at;
    }
}

// ---------------------------------------------------------------------------
// K2 fed by the MOG kernel's 1-bit tiles (hk_mog_apply_batch_tiles): per
// 32 x 4 pixel tile and frame four words -- the foreground bits of the tile's
// even pixels (16 per row: rows 0-1, rows 2-3) and of its odd pixels.  One
// lane per tile interleaves them into four words of the padded 1-bit image
// and finds the candidate border starts of its 128 pixels on whole words; the
// neighbour bits it needs from the tiles around come straight out of their
// planes (bit 31 of a word = bit 15 of the odd plane, bit 0 = bit 0 of the
// even plane).  1/8 of the bytes blob_scan_k reads, and no byte masks at all.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned spread16(unsigned v) {  // bit i -> bit 2i
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__device__ __forceinline__ unsigned plane_row(unsigned p01, unsigned p23,
                                              int r) {
    return ((r & 2 ? p23 : p01) >> (16 * (r & 1))) & 0xFFFFu;
}

__device__ __forceinline__ void emit_candidates(
    unsigned outer, unsigned hole, int f, int y, int x0, Cand *cands,
    unsigned *ctr, int band_rows, int nbands, unsigned band_cap, int ncols,
    unsigned *active) {
    const unsigned c = __popc(outer) + __popc(hole);
    if (c == 0) return;
    const unsigned bucket =
        ((unsigned)f * nbands + (unsigned)(y / band_rows)) * ncols +
        (unsigned)(x0 / (32 * COL_WORDS));
    unsigned at = atomicAdd(&ctr[CTR_WORDS + bucket], c);
    if (at == 0u)       // the bucket's first candidates: list it for tracing
        active[atomicAdd(&ctr[CTR_NACTIVE], 1u)] = bucket;
    if (at + c > band_cap) atomicOr(&ctr[CTR_FLAGS], FLAG_BAND_FULL);
    Cand *list = cands + (size_t)bucket * band_cap;
    while (outer) {
        const int i = __ffs(outer) - 1;
        outer &= outer - 1;
        if (at < band_cap)
            list[at] = Cand{((unsigned)y << 16) | (unsigned)(x0 + i),
                            (unsigned)f << 1};
        ++at;
    }
    while (hole) {
        const int i = __ffs(hole) - 1;
        hole &= hole - 1;
        if (at < band_cap)      // origin = the foreground pixel left of it
            list[at] = Cand{((unsigned)y << 16) | (unsigned)(x0 + i - 1),
                            ((unsigned)f << 1) | 1u};
        ++at;
    }
}

__global__ __launch_bounds__(256) void blob_scan_tiles_k(
    const uint4 *__restrict__ tiles, int tiles_x, int tiles_y, int nframes,
    int H, uint32_t *__restrict__ bits, int words_per_row,
    Cand *__restrict__ cands, unsigned *__restrict__ ctr, int band_rows,
    int nbands, unsigned band_cap, int ncols, unsigned *__restrict__ active) {
    const long per_frame = (long)tiles_x * tiles_y;
    const long total = (long)nframes * per_frame;
    for (long tid = (long)blockIdx.x * 256 + threadIdx.x; tid < total;
         tid += (long)gridDim.x * 256) {
        const int f = (int)(tid / per_frame);
        const int rest = (int)(tid - (long)f * per_frame);
        const int ty = rest / tiles_x, tx = rest - ty * tiles_x;
        const uint4 *frame = tiles + (long)f * per_frame;
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        // x = even rows 0-1, y = even rows 2-3, z = odd rows 0-1, w = odd 2-3
        const uint4 me = frame[rest];
        const uint4 lt = tx > 0 ? frame[rest - 1] : zero;
        const uint4 rt = tx + 1 < tiles_x ? frame[rest + 1] : zero;
        uint4 up = zero, ul = zero, ur = zero;
        if (ty > 0) {
            up = frame[rest - tiles_x];
            if (tx > 0) ul = frame[rest - tiles_x - 1];
            if (tx + 1 < tiles_x) ur = frame[rest - tiles_x + 1];
        }
        // the row above the tile: its word, and the bits beside it
        unsigned n = spread16(up.y >> 16) | (spread16(up.w >> 16) << 1);
        unsigned n_left31 = ul.w >> 31;
        unsigned n_right0 = (ur.y >> 16) & 1u;
        uint32_t *dst = bits + ((long)f * (H + 2) + ty * 4 + 1) *
                                   words_per_row + 1 + tx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned cur = spread16(plane_row(me.x, me.y, r)) |
                                 (spread16(plane_row(me.z, me.w, r)) << 1);
            const unsigned left31 = plane_row(lt.z, lt.w, r) >> 15;
            const unsigned right0 = plane_row(rt.x, rt.y, r) & 1u;
            dst[(long)r * words_per_row] = cur;
            const unsigned w = (cur << 1) | left31;
            const unsigned nw = (n << 1) | n_left31;
            const unsigned ne = (n >> 1) | (n_right0 << 31);
            unsigned outer = cur & ~w & ~n & ~nw & ~ne;
            unsigned hole = ~cur & w & n;
            if (outer | hole) {     // few rows of few tiles: the words to the
                                    // right of this row and of the one above
                const unsigned cur_r =
                    spread16(plane_row(rt.x, rt.y, r)) |
                    (spread16(plane_row(rt.z, rt.w, r)) << 1);
                const unsigned n_r =
                    r == 0 ? spread16(ur.y >> 16) | (spread16(ur.w >> 16) << 1)
                           : spread16(plane_row(rt.x, rt.y, r - 1)) |
                                 (spread16(plane_row(rt.z, rt.w, r - 1)) << 1);
                prune_candidates(outer, hole,
                                 cur | ((unsigned long long)cur_r << 32),
                                 n | ((unsigned long long)n_r << 32), 64);
            }
            emit_candidates(outer, hole, f, ty * 4 + r, tx * 32, cands, ctr,
                            band_rows, nbands, band_cap, ncols, active);
            n = cur;
            n_left31 = left31;
            n_right0 = right0;
        }
    }
}

// ---------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------
// The same picture as 1 bit per pixel (written by blob_scan_k); `rows` is
// the LDS copy of a band of it: all columns (one zero guard word on each
// side), image rows row0 .. row0 + nrows - 1.  Every step of a border is a
// dependent read of the 3x3 neighbourhood: from the byte image in HBM that is
// 8 scattered loads and a memory round trip per step, here six LDS reads.
struct BandImg {
    const uint32_t *rows;   // LDS: WIN_WORDS words per row
    int w0, row0;           // first word (padded frame) and image row held
    const uint32_t *frame;  // the padded frame in global memory (for fg())
    int wpr, W, H;

    __device__ __forceinline__ unsigned fg(int x, int y) const {
        if (x < 0 || x >= W || y < 0 || y >= H) return 0u;
        return (frame[(long)(y + 1) * wpr + 1 + (x >> 5)] >> (x & 31)) & 1u;
    }
    // bits x-1, x, x+1 of image row y
    __device__ __forceinline__ unsigned three(int x, int y) const {
        const uint32_t *q =
            rows + (y - row0) * WIN_WORDS + 1 + ((x - 1) >> 5) - w0;
        const unsigned long long v =
            (unsigned long long)q[0] | ((unsigned long long)q[1] << 32);
        return (unsigned)(v >> ((x - 1) & 31)) & 7u;
    }
    __device__ __forceinline__ unsigned ring(int x, int y) const {
        const unsigned u = three(x, y - 1), m = three(x, y),
                       d = three(x, y + 1);
        // E NE N NW W SW S SE
        return (m >> 2) | ((u >> 2) << 1) | (((u >> 1) & 1u) << 2) |
               ((u & 1u) << 3) | ((m & 1u) << 4) | ((d & 1u) << 5) |
               (((d >> 1) & 1u) << 6) | ((d >> 2) << 7);
    }
};

struct Trace {
    long long a00, a10, a01;
    int ns, nd, npts;
    bool valid;
};

__device__ __forceinline__ bool before(int x, int y, int rx, int ry) {
    return y < ry || (y == ry && x < rx);
}

// Bit d set iff the neighbour of (x, y) in direction d lies before (rx, ry)
// in raster order (E NE N NW W SW S SE): before() for all 8 at once.
__device__ __forceinline__ unsigned before_ring(int x, int y, int rx, int ry) {
    const unsigned lm = x - 1 < rx, l0 = x < rx, lp = x + 1 < rx;
    // pixels x+1, x, x-1 of a row, as direction bits of the row above
    // (NE N NW = bits 1 2 3), the same row (E W = 0 4), the row below
    // (SE S SW = 7 6 5)
    const int dy = y - ry;
    const unsigned up = dy <= 0 ? 0x0Eu
                      : dy == 1 ? (lp << 1) | (l0 << 2) | (lm << 3) : 0u;
    const unsigned mid = dy < 0 ? 0x11u : dy == 0 ? lp | (lm << 4) : 0u;
    const unsigned dn = dy < -1 ? 0xE0u
                      : dy == -1 ? (lp << 7) | (l0 << 6) | (lm << 5) : 0u;
    return up | mid | dn;
}

// Border following from the foreground pixel (x0, y0), icvFetchContour order.
// `hole` selects the start direction.  A trace is valid iff it closes within
// max_pts points and (outer) visits no pixel before (x0, y0) in raster order
// / (hole) examines no background pixel before (x0 + 1, y0).  A border that
// reaches a row more than max_pts / 2 below its start cannot close within
// max_pts points (a step changes the row by at most one), so it is given up
// there: no pixel below row y0 + max_pts / 2 + 1 is ever read (and none above
// row y0 - 2: see the checks after each step).
template <bool WRITE, class Image>
__device__ Trace follow_border(const Image &img, int x0, int y0, bool hole,
                               int max_pts, ushort2 *__restrict__ pts) {
    Trace t{0, 0, 0, 0, 0, 0, false};
    const int rx = hole ? x0 + 1 : x0, ry = y0;
    const int s0 = hole ? 0 : 4;

    unsigned nb = img.ring(x0, y0);
    if (nb == 0u) {                     // single pixel domain
        if (WRITE) pts[0] = make_ushort2((unsigned short)x0, (unsigned short)y0);
        t.npts = 1;
        t.valid = !hole;
        return t;
    }
    // first neighbour clockwise from s0 (directions s0-1, s0-2, ...)
    // (s0 = 0 for a hole: directions 7, 6, ... 1 are examined in turn)
    int s = s0;
    for (int j = 1; j < 8; ++j) {
        s = (s0 - j) & 7;
        if ((nb >> s) & 1u) break;
    }
    if (hole) {             // background pixels examined: directions s+1 .. 7
        const unsigned seen = (0xFFu << (s + 1)) & 0xFFu;
        if (before_ring(x0, y0, rx, ry) & seen) return t;
    }
    const int x1 = x0 + dir_dx(s), y1 = y0 + dir_dy(s);

    int x3 = x0, y3 = y0;
    for (;;) {
        // sweep counter-clockwise from s + 1 to the next foreground pixel
        const int first = (s + 1) & 7;
        const unsigned rot = ((nb | (nb << 8)) >> first) & 0xFFu;
        const int skip = __ffs(rot) - 1;            // rot != 0 always
        if (hole) {         // background pixels examined: first .. first+skip-1
            const unsigned run = (1u << skip) - 1u;
            const unsigned seen = ((run << first) | (run << first >> 8)) & 0xFFu;
            if (before_ring(x3, y3, rx, ry) & seen) return t;
        }
        s = (first + skip) & 7;
        const int x4 = x3 + dir_dx(s), y4 = y3 + dir_dy(s);

        if (t.npts >= max_pts) return t;
        if (WRITE) pts[t.npts] = make_ushort2((unsigned short)x3, (unsigned short)y3);
        ++t.npts;
        if (s & 1) ++t.nd; else ++t.ns;
        // x3*y4 - x4*y3 with (x4, y4) = (x3, y3) + step: small, 32-bit exact
        const int dxy = x3 * dir_dy(s) - dir_dx(s) * y3;
        t.a00 += dxy;
        t.a10 += (long long)dxy * (x3 + x4);
        t.a01 += (long long)dxy * (y3 + y4);

        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4;
        y3 = y4;
        // An outer border that reaches a pixel before its start in raster
        // order is not traced from its first pixel.  A hole's pixels all lie
        // in rows >= y0 if (x0 + 1, y0) is its first, and its border within
        // one pixel of them: a hole border that climbs above row y0 - 1
        // belongs to a hole that starts further up.  Either way the candidate
        // retires here, before the neighbourhood of (x3, y3) is read: no
        // pixel above row y0 - 2 is ever read.
        if (hole ? y3 < y0 - 1 : before(x3, y3, rx, ry)) return t;
        if (y3 - y0 > max_pts / 2) return t;
        if (x3 - x0 > max_pts / 2 || x0 - x3 > max_pts / 2) return t;
        s = (s + 4) & 7;
        nb = img.ring(x3, y3);
    }
    t.valid = true;
    return t;
}

// ---------------------------------------------------------------------------
// The same border following, written for the tracing kernel's LDS window and
// for as few instructions per step as possible (a border is a chain of
// dependent steps on one lane: its length times the instructions of a step is
// what the kernel's duration is made of).
//  * one LDS address per step, three ds_read2_b32 off it, three alignbits;
//  * positions relative to the start, packed (Y + 256) << 16 | (X + 256): the
//    closing test is two compares, "before the start in raster order" (outer)
//    / "above row y0 - 1" (hole) / "too far below" one unsigned range test;
//  * Green sums in 32 bits on the relative coordinates (|X|, |Y| <= 128, at
//    most 256 steps) and moved to image coordinates at the end with the exact
//    identities  a00 = a00',  a10 = a10' + 3 x0 a00',  a01 = a01' + 3 y0 a00'
//    (closed borders: sum dx = sum dy = 0, sum dx (X3 + X4) = 0,
//    sum dy (X3 + X4) = a00', sum dx (Y3 + Y4) = -a00');
//  * the points are written while tracing (into a first-pass slot), so a
//    border that passes the filters is not followed a second time;
//  * one exit per iteration.
// Same decisions and same sums as follow_border(), which stays as the
// reference form (and as the fallback when the first-pass slots run out).
struct LeanTrace {
    int a00, a10, a01, nd, npts;
    bool valid;
};

__device__ __forceinline__ unsigned window_ring(const uint32_t *lds, int U,
                                                int YR) {
    // U = bit offset of pixel x - 1 from the window's first bit, YR = row of
    // the window that holds image row y
    const uint32_t *q = lds + (YR - 1) * WIN_WORDS + (U >> 5);
    const unsigned u = __builtin_amdgcn_alignbit(q[1], q[0], U);
    const unsigned m =
        __builtin_amdgcn_alignbit(q[WIN_WORDS + 1], q[WIN_WORDS], U);
    const unsigned d =
        __builtin_amdgcn_alignbit(q[2 * WIN_WORDS + 1], q[2 * WIN_WORDS], U);
    // E NE N NW W SW S SE = m2 u2 u1 u0 m0 d0 d1 d2
    const unsigned urev = __builtin_bitreverse32(u) >> 29;
    return ((m >> 2) & 1u) | (urev << 1) | ((m & 1u) << 4) | ((d & 7u) << 5);
}

template <bool HOLES>
__device__ __forceinline__ LeanTrace trace_window(
    const uint32_t *lds, int U0, int YR0, int x0, int y0, bool hole,
    int max_pts, uint32_t *__restrict__ out) {
    LeanTrace t{0, 0, 0, 0, 0, false};
    const unsigned PK0 = (256u << 16) | 256u;
    const unsigned absk = (((unsigned)y0 << 16) | (unsigned)x0) - PK0;
    unsigned nb = window_ring(lds, U0, YR0);
    if (nb == 0u) {                     // single pixel domain
        if (out) out[0] = PK0 + absk;
        t.npts = 1;
        t.valid = !hole;
        return t;
    }
    const int s0 = (HOLES && hole) ? 0 : 4;
    // first neighbour clockwise from s0: directions s0-1, s0-2, ... = the
    // highest set bit of the ring rotated so that direction s0 + k is bit k
    const unsigned r0 = ((nb | (nb << 8)) >> s0) & 0xFFu;
    int s = (s0 + (31 - __clz((int)r0))) & 7;
    const int rx = x0 + 1;
    if (HOLES && hole) {    // background pixels examined: directions s+1 .. 7
        const unsigned seen = (0xFFu << (s + 1)) & 0xFFu;
        if (before_ring(x0, y0, rx, y0) & seen) return t;
    }
    const unsigned PK1 = PK0 + (unsigned)(dir_dy(s) * 65536 + dir_dx(s));
    const int half = max_pts / 2;
    // positions a trace may be at: LO <= packed < HI, |X| <= half
    const unsigned LO = (HOLES && hole) ? (255u << 16) : PK0;
    const unsigned SPAN = ((256u + (unsigned)half + 1u) << 16) - LO;

    int X = 0, Y = 0, npts = 0, nd = 0, a00 = 0, a10 = 0, a01 = 0;
    unsigned pk3 = PK0;
    bool valid = false;
    for (;;) {
        // sweep counter-clockwise from s + 1 to the next foreground pixel
        const int first = (s + 1) & 7;
        const unsigned rot = ((nb | (nb << 8)) >> first) & 0xFFu;
        const int skip = __ffs((int)rot) - 1;           // rot != 0 always
        bool bad = npts >= max_pts;
        if (HOLES) {        // background pixels examined: first .. first+skip-1
            const unsigned run = (1u << skip) - 1u;
            const unsigned seen =
                ((run << first) | (run << first >> 8)) & 0xFFu;
            bad |= hole && (before_ring(x0 + X, y0 + Y, rx, y0) & seen) != 0u;
        }
        s = (first + skip) & 7;
        if (out && !bad) out[npts] = pk3 + absk;
        ++npts;
        nd += s & 1;
        const int dx = dir_dx(s), dy = dir_dy(s);
        const int dxy = __mul24(X, dy) - __mul24(dx, Y);
        a00 += dxy;
        a10 += __mul24(dxy, 2 * X + dx);
        a01 += __mul24(dxy, 2 * Y + dy);
        const unsigned pk4 = pk3 + (unsigned)(dy * 65536 + dx);
        const bool closed = pk4 == PK0 && pk3 == PK1;
        X += dx;
        Y += dy;
        pk3 = pk4;
        // an outer border that reaches a pixel before its start, a hole
        // border above row y0 - 1, any border too far below or beside its
        // start: given up before the neighbourhood there is read
        const bool away = pk3 - LO >= SPAN ||
                          (unsigned)(X + half) > 2u * (unsigned)half;
        valid = closed && !bad;
        if (bad || closed || away) break;
        s = (s + 4) & 7;
        nb = window_ring(lds, U0 + X, YR0 + Y);
    }
    t.a00 = a00;
    t.a10 = a10;
    t.a01 = a01;
    t.nd = nd;
    t.npts = npts;
    t.valid = valid;
    return t;
}

// The candidates of one (frame, band, block of columns).
__device__ __forceinline__ void trace_bucket(
    const unsigned bucket, uint32_t *band_lds,
    const uint32_t *__restrict__ bits, int words_per_row, int W,
    int H, int band_rows, int nbands, const Cand *__restrict__ cands,
    unsigned *__restrict__ ctr, BlobRec *__restrict__ recs,
    ushort2 *__restrict__ pts, unsigned cap_blob, const DevParams &prm,
    unsigned band_cap, int ncols, uint32_t *__restrict__ tmp_pts,
    unsigned cap_tmp) {
    unsigned n = ctr[CTR_WORDS + bucket];
    if (n == 0) return;                     // workgroup-uniform
    if (n > band_cap) n = band_cap;
    const int col = (int)(bucket % (unsigned)ncols);
    const unsigned fb = bucket / (unsigned)ncols;
    const int frame = (int)(fb / (unsigned)nbands);
    // The window: image rows ymin - 2 .. ymax + max_pts/2 + 2 for the first
    // and last start row among the candidates (a border that starts on row y0
    // steps up to row y0 - 1 at most, and the neighbourhood read there
    // reaches row y0 - 2; it is given up max_pts/2 rows below y0), words w0 ..
    // w0 + WIN_WORDS - 1 of the padded rows; zeros outside the padded frame.
    int ymin = 0x7FFFFFFF, ymax = 0;
    for (unsigned i = threadIdx.x; i < n; i += TRACE_THREADS) {
        const int y = (int)(cands[(size_t)bucket * band_cap + i].pos >> 16);
        ymin = y < ymin ? y : ymin;
        ymax = y > ymax ? y : ymax;
    }
    for (int d = 32; d > 0; d >>= 1) {
        const int a = __shfl_xor(ymin, d), c = __shfl_xor(ymax, d);
        ymin = a < ymin ? a : ymin;
        ymax = c > ymax ? c : ymax;
    }
    const int row0 = ymin - 2;
    int nrows = ymax - ymin + prm.max_pts / 2 + 5;
    if (row0 + nrows > H + 1) nrows = H + 1 - row0;
    const int w0 = col * COL_WORDS - 4;
    const uint32_t *fbits = bits + (long)frame * (H + 2) * words_per_row;
    {
        // 16-byte groups, eight loads in flight per lane before the stores
        constexpr int G = WIN_WORDS / 4;
        const int total = nrows * G;
        uint4 *lds4 = reinterpret_cast<uint4 *>(band_lds);
        for (int k0 = 0; k0 < total; k0 += 8 * TRACE_THREADS) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * TRACE_THREADS + (int)threadIdx.x;
                const int r = k / G, w = w0 + 4 * (k - r * G);
                const int prow = row0 + r + 1;      // row of the padded frame
                v[j] = make_uint4(0u, 0u, 0u, 0u);
                if (k < total && prow >= 0 && prow < H + 2 && w >= 0 &&
                    w < words_per_row)
                    v[j] = *reinterpret_cast<const uint4 *>(
                        fbits + (long)prow * words_per_row + w);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * TRACE_THREADS + (int)threadIdx.x;
                if (k < total) lds4[k] = v[j];
            }
        }
    }
    __syncthreads();
    const BandImg img{band_lds, w0, row0, fbits, words_per_row, W, H};

    // first-pass point lists: one slot per candidate of this workgroup
    unsigned tmp_base = 0;
    if (threadIdx.x == 0) tmp_base = atomicAdd(&ctr[CTR_NTMP], n);
    tmp_base = __shfl(tmp_base, 0);

    for (unsigned i0 = 0; i0 < n; i0 += TRACE_THREADS) {
        const unsigned i = i0 + threadIdx.x;
        const bool have = i < n;
        const Cand c = have ? cands[(size_t)bucket * band_cap + i]
                            : Cand{0u, 0u};
        const int x0 = (int)(c.pos & 0xFFFFu), y0 = (int)(c.pos >> 16);
        const bool hole = (c.fh & 1u) != 0u;
        const bool wave_holes = __any(have && hole) != 0;   // wave-uniform
        if (!have) continue;
        const unsigned tslot = tmp_base + i;
        uint32_t *out = tslot < cap_tmp ? tmp_pts + (size_t)tslot * MAXPTS
                                        : nullptr;
        const int U0 = x0 - 1 + 32 * (1 - w0), YR0 = y0 - row0;
        const LeanTrace lt =
            wave_holes ? trace_window<true>(band_lds, U0, YR0, x0, y0, hole,
                                            prm.max_pts, out)
                       : trace_window<false>(band_lds, U0, YR0, x0, y0, false,
                                             prm.max_pts, out);
        if (!lt.valid) continue;
        Trace t;
        t.a00 = lt.a00;
        t.a10 = (long long)lt.a10 + 3LL * x0 * lt.a00;
        t.a01 = (long long)lt.a01 + 3LL * y0 * lt.a00;
        t.nd = lt.nd;
        t.ns = lt.npts - lt.nd;
        t.npts = lt.npts;

        // contourMoments(): m00, m10, m01 from the Green sums
        double m00 = 0.0, m10 = 0.0, m01 = 0.0;
        if (t.a00 != 0) {
            const double h = t.a00 > 0 ? 0.5 : -0.5;
            const double sixth = t.a00 > 0
                                     ? 0.16666666666666666666666666666667
                                     : -0.16666666666666666666666666666667;
            m00 = (double)t.a00 * h;
            m10 = (double)t.a10 * sixth;
            m01 = (double)t.a01 * sixth;
        }
        const double area = m00;
        if (area < prm.minArea || area >= prm.maxArea) continue;
        // arcLength: every segment is 1.0f or sqrtf(2.0f), summed in double
        const double perimeter =
            (double)t.ns + (double)t.nd * (double)1.41421354f;
        const double ratio =
            4 * 3.1415926535897932384626433832795 * area /
            (perimeter * perimeter);
        if (ratio < prm.minCirc || ratio >= prm.maxCirc) continue;
        if (m00 == 0.0) continue;
        const double cx = m10 / m00, cy = m01 / m00;
        if (prm.filterByColor) {
            const int px = __double2int_rn(cx), py = __double2int_rn(cy);
            const int v = img.fg(px, py) ? 255 : 0;
            if (v != prm.blobColor) continue;
        }

        const unsigned slot = atomicAdd(&ctr[CTR_NBLOB], 1u);
        if (slot >= cap_blob) continue;         // host reports the overflow
        BlobRec r;
        r.frame = frame;
        r.key = ((unsigned)y0 << 16) | (unsigned)(hole ? x0 + 1 : x0);
        r.is_hole = hole ? 1 : 0;
        r.npts = t.npts;
        r.pts_at = out ? (int)tslot : -1;
        r.pad_ = 0;
        r.a00 = t.a00;
        r.a10 = t.a10;
        r.a01 = t.a01;
        r.cx = cx;
        r.cy = cy;
        r.d2lo = r.d2hi = 0.0;
        recs[slot] = r;
        if (!out)       // no first-pass slot was left: follow it once more
            (void)follow_border<true>(img, x0, y0, hole, prm.max_pts,
                                      pts + (size_t)slot * MAXPTS);
    }
}

// A fixed number of single-wave workgroups take the buckets that hold
// candidates (listed by the scan kernel) off a queue until it is empty.  One
// workgroup per bucket of the batch -- 17 000 per 128 frames at 1080p, two
// thirds of them empty -- made every one of them wait for a wave slot and
// 14 KB of LDS among the MOG waves of the next chunk just to find that out.
__global__ __launch_bounds__(TRACE_THREADS) void blob_trace_k(
    const uint32_t *__restrict__ bits, int words_per_row, int W,
    int H, int band_rows, int nbands, const Cand *__restrict__ cands,
    unsigned *__restrict__ ctr, BlobRec *__restrict__ recs,
    ushort2 *__restrict__ pts, unsigned cap_blob, const DevParams prm,
    unsigned band_cap, int ncols, uint32_t *__restrict__ tmp_pts,
    unsigned cap_tmp, const unsigned *__restrict__ active) {
    extern __shared__ uint32_t band_lds[];
    const unsigned n_active = ctr[CTR_NACTIVE];
    for (;;) {
        unsigned at = 0;
        if (threadIdx.x == 0) at = atomicAdd(&ctr[CTR_NEXT], 1u);
        at = __shfl(at, 0);
        if (at >= n_active) break;          // every wave gets here
        trace_bucket(active[at], band_lds, bits, words_per_row, W, H,
                     band_rows, nbands, cands, ctr, recs, pts, cap_blob, prm,
                     band_cap, ncols, tmp_pts, cap_tmp);
        __syncthreads();                    // before the window is refilled
    }
}

// ---------------------------------------------------------------------------
// K3b
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void blob_radius_k(
    const unsigned *__restrict__ ctr, unsigned cap_blob,
    BlobRec *__restrict__ recs, const ushort2 *__restrict__ pts,
    const ushort2 *__restrict__ tmp_pts, HostRec *__restrict__ hrecs,
    unsigned *__restrict__ h_ctr, unsigned *__restrict__ next_ctr,
    unsigned next_words) {
    __shared__ double d2[MAXPTS];
    __shared__ double mid[2];
    // The counters of this handle's next batch (its other set) start at
    // zero: cleared here, by the last kernel of this batch, so that the next
    // batch does not begin with a fill kernel that has to find wave slots
    // among the MOG waves (160 us per chunk in a pipelined batch).
    for (unsigned i = blockIdx.x * 64u + threadIdx.x; i < next_words;
         i += gridDim.x * 64u)
        next_ctr[i] = 0u;
    // hrecs and h_ctr are pinned host memory: what the host reads goes there
    // straight from this kernel (no copy kernels behind it, which would queue
    // for a slot beside the next chunk's MOG launch: 150 us per chunk).  The
    // counters are final before this kernel starts.
    if (blockIdx.x == 0 && threadIdx.x < CTR_WORDS)
        h_ctr[threadIdx.x] = ctr[threadIdx.x];
    unsigned n = ctr[CTR_NBLOB];
    if (n > cap_blob) n = cap_blob;
    const int lane = threadIdx.x;
    for (unsigned slot = blockIdx.x; slot < n; slot += gridDim.x) {
        const int m = recs[slot].npts;
        const double cx = recs[slot].cx, cy = recs[slot].cy;
        const int at = recs[slot].pts_at;
        const ushort2 *src = at >= 0 ? tmp_pts + (size_t)at * MAXPTS
                                     : pts + (size_t)slot * MAXPTS;
        for (int i = lane; i < m; i += 64) {
            const ushort2 p = src[i];
            const double ddx = cx - (double)p.x;
            const double ddy = cy - (double)p.y;
            d2[i] = ddx * ddx + ddy * ddy;
        }
        __syncthreads();
        const int lo = (m - 1) / 2, hi = m / 2;
        for (int i = lane; i < m; i += 64) {
            const double v = d2[i];
            int rank = 0;
            for (int j = 0; j < m; ++j) {
                const double u = d2[j];
                rank += (u < v) || (u == v && j < i);
            }
            if (rank == lo) mid[0] = v;
            if (rank == hi) mid[1] = v;
        }
        __syncthreads();
        if (lane == 0) {
            // sort + middle of OpenCV's dists vector: (dists[(n-1)/2] +
            // dists[n/2]) / 2 with dists = the roots (IEEE sqrt, monotonic)
            recs[slot].d2lo = mid[0];
            recs[slot].d2hi = mid[1];
            HostRec h;
            h.key = recs[slot].key;
            h.frame = recs[slot].frame;
            h.cx = cx;
            h.cy = cy;
            h.radius = (sqrt(mid[0]) + sqrt(mid[1])) / 2.;
            hrecs[slot] = h;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K4 (host)
// ---------------------------------------------------------------------------
struct Center {
    double x, y, radius;
};

// SimpleBlobDetector::detect()'s loop over thresholds, given that every
// threshold sees the same centres `cur` (binary input).  Literal replay;
// returns the groups in creation order.
void replay_groups(const std::vector<Center> &cur, int n_thresholds,
                   const hk_blob_params &p,
                   std::vector<std::vector<Center>> &groups) {
    const double min_dist = p.minDistBetweenBlobs;
    for (int pass = 0; pass < n_thresholds; ++pass) {
        const size_t existing = groups.size();
        std::vector<std::vector<Center>> fresh;
        for (const Center &c : cur) {
            bool is_new = true;
            for (size_t j = 0; j < existing; ++j) {
                std::vector<Center> &g = groups[j];
                const Center &mid = g[g.size() / 2];
                const double ddx = mid.x - c.x, ddy = mid.y - c.y;
                const double dist = std::sqrt(ddx * ddx + ddy * ddy);
                is_new = dist >= min_dist && dist >= mid.radius &&
                         dist >= c.radius;
                if (!is_new) {
                    // insert keeping the group sorted by radius
                    size_t k = g.size();
                    g.push_back(c);
                    while (k > 0 && c.radius < g[k - 1].radius) {
                        g[k] = g[k - 1];
                        --k;
                    }
                    g[k] = c;
                    break;
                }
            }
            if (is_new) fresh.push_back(std::vector<Center>(1, c));
        }
        for (auto &g : fresh) groups.push_back(std::move(g));
    }
}

// Keypoint of a group: location = sum of the members' centres (sequential
// double adds) times 1/count, cast to float; size = twice the middle radius.
void emit_group(const Center *g, size_t n, std::vector<float> &out) {
    double sx = 0, sy = 0, normalizer = 0;
    for (size_t i = 0; i < n; ++i) {
        sx += 1.0 * g[i].x;
        sy += 1.0 * g[i].y;
        normalizer += 1.0;
    }
    const double inv = 1. / normalizer;
    sx *= inv;
    sy *= inv;
    out.push_back((float)sx);
    out.push_back((float)sy);
    out.push_back((float)(g[n / 2].radius) * 2.0f);
}

// The grouping loop without the 17-fold repetition.  A centre that no other
// centre can ever be matched with (farther than minDistBetweenBlobs and than
// both radii from every other centre; every group's middle element is a copy
// of some centre) ends as a group of n_thresholds copies of itself, created
// in the first pass at its own position in the group list.  Only the
// remaining centres go through the literal replay, among themselves; groups
// the replay creates after the first pass come last, as in the full loop.
void group_keypoints(const Center *cur, size_t n, int n_thresholds,
                     const hk_blob_params &p, std::vector<float> &out) {
    if (n == 0 || n_thresholds <= 0) return;
    const double min_dist = p.minDistBetweenBlobs;
    // (a detection thread calls this for every frame: the scratch vectors
    // live across calls)
    static thread_local std::vector<char> linked;
    static thread_local std::vector<int> by_y;
    linked.assign(n, 0);
    bool any = false;
    // Two centres can only be matched if they are closer than the largest of
    // minDistBetweenBlobs and their radii: with the centres in order of y
    // (a few dozen per frame: insertion sort) each one is tested against
    // those within that bound below it instead of against all the others.
    // Which pairs are tested does not matter to the outcome, only whether a
    // centre has any partner.
    double bound = min_dist;
    by_y.resize(n);
    for (size_t i = 0; i < n; ++i) {
        if (cur[i].radius > bound) bound = cur[i].radius;
        int k = (int)i;
        while (k > 0 && cur[(size_t)by_y[(size_t)k - 1]].y > cur[i].y) {
            by_y[(size_t)k] = by_y[(size_t)k - 1];
            --k;
        }
        by_y[(size_t)k] = (int)i;
    }
    const double reach = bound * (1 + 1e-9);
    for (size_t a = 0; a < n; ++a)
        for (size_t b = a + 1; b < n; ++b) {
            const size_t i = (size_t)by_y[a], j = (size_t)by_y[b];
            // (centres are quotients of finite sums with m00 != 0: finite)
            if (cur[j].y - cur[i].y > reach) break;
            const double ddx = cur[i].x - cur[j].x, ddy = cur[i].y - cur[j].y;
            const double q = ddx * ddx + ddy * ddy;
            // "dist >= min_dist, radius i and radius j" with dist = sqrt(q):
            // decided on the squares unless q is within rounding of the
            // largest bound's square (sqrt is monotonic, correctly rounded)
            double m = min_dist;
            if (cur[i].radius > m) m = cur[i].radius;
            if (cur[j].radius > m) m = cur[j].radius;
            const double m2 = m * m;
            bool apart;
            if (q > m2 * (1 + 1e-12)) {
                apart = true;
            } else if (q < m2 * (1 - 1e-12)) {
                apart = false;
            } else {
                const double dist = std::sqrt(q);
                apart = dist >= min_dist && dist >= cur[i].radius &&
                        dist >= cur[j].radius;
            }
            if (!apart) {
                linked[i] = linked[j] = 1;
                any = true;
            }
        }
    std::vector<std::vector<Center>> groups;
    size_t first_pass = 0;
    if (any) {
        std::vector<Center> sub;
        for (size_t i = 0; i < n; ++i)
            if (linked[i]) sub.push_back(cur[i]);
        first_pass = sub.size();
        replay_groups(sub, n_thresholds, p, groups);
    }
    // An isolated centre ends as a group of n_thresholds copies of itself:
    // the same sequential sums as emit_group() makes, for all centres of the
    // frame side by side (the additions of different centres do not depend
    // on each other: the compiler vectorises the inner loops)
    static thread_local std::vector<double> sums;
    sums.resize(4 * n);
    double *const sx = sums.data(), *const sy = sx + n, *const cx = sy + n,
                 *const cy = cx + n;
    double normalizer = 0;
    for (size_t i = 0; i < n; ++i) {
        cx[i] = 1.0 * cur[i].x;
        cy[i] = 1.0 * cur[i].y;
        sx[i] = sy[i] = 0;
    }
    for (int k = 0; k < n_thresholds; ++k) {
        for (size_t i = 0; i < n; ++i) sx[i] += cx[i];
        for (size_t i = 0; i < n; ++i) sy[i] += cy[i];
        normalizer += 1.0;
    }
    const double inv = 1. / normalizer;
    size_t next_sub = 0;
    for (size_t i = 0; i < n; ++i) {
        if (linked[i]) {
            const std::vector<Center> &g = groups[next_sub++];
            if ((int)g.size() >= p.minRepeatability)
                emit_group(g.data(), g.size(), out);
        } else if (n_thresholds >= p.minRepeatability) {
            out.push_back((float)(sx[i] * inv));
            out.push_back((float)(sy[i] * inv));
            out.push_back((float)(cur[i].radius) * 2.0f);
        }
    }
    for (size_t k = first_pass; k < groups.size(); ++k)
        if ((int)groups[k].size() >= p.minRepeatability)
            emit_group(groups[k].data(), groups[k].size(), out);
}

// The contours of a batch that passed the filters: centres of all frames,
// frame f's in [off[f], off[f + 1]), in the order findContours(RETR_LIST)
// lists them.
struct Contours {
    std::vector<Center> centers;
    std::vector<int> off;
    void swap(Contours &o) {
        centers.swap(o.centers);
        off.swap(o.off);
    }
    size_t frames() const { return off.empty() ? 0 : off.size() - 1; }
};

}  // namespace

struct hk_blob {
    int width = 0, height = 0, max_frames = 0, device = 0;
    hk_blob_params params;
    DevParams dev;
    int n_thresholds = 0;
    unsigned cap_blob = 0;
    int band_rows = 0, nbands = 0;      // K3 works on bands of rows,
    int ncols = 0;                      // in blocks of COL_WORDS words
    size_t trace_lds = 0;               // bytes of LDS a K3 workgroup needs
    Cand *cands = nullptr;
    unsigned *ctr = nullptr;
    BlobRec *recs = nullptr;
    ushort2 *pts = nullptr;
    uint32_t *tmp_pts = nullptr;        // first-pass point lists, cap_tmp slots
    unsigned cap_tmp = 0;
    unsigned *active = nullptr;         // buckets that hold candidates
    uint32_t *bits = nullptr;           // [max_frames][H + 2][words_per_row]
    int words_per_row = 0;              // (W + 31) / 32 + 2, rounded up to 4
    unsigned *h_ctr = nullptr;          // pinned
    HostRec *h_recs = nullptr;          // pinned, cap_blob entries; the
    HostRec *hrecs = nullptr;           // same memory as the device sees it,
    unsigned *h_ctr_dev = nullptr;      // and h_ctr likewise
    // two sets of counters: `ctr` is the one the current (or last) batch
    // uses, ctr_set[1 - ctr_at] the next batch's, cleared by this batch's
    // radius kernel (ctr_clean) or, failing that, by a memset
    unsigned *ctr_set[2] = {nullptr, nullptr};
    bool ctr_clean[2] = {false, false};
    int ctr_at = 0;
    size_t ctr_words = 0;
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;
    int pending_frames = 0;             // batch between enqueue and finish
    unsigned band_cap = BAND_CAP0;      // entries per (frame, band) in cands
    // the batch in flight (collect() runs it again with larger buffers if
    // a list overflowed)
    const uint8_t *masks = nullptr;     // null: the batch is in `tiles`
    long frame_stride = 0, pitch = 0;
    uint32_t *tiles = nullptr;          // [max_frames][W/32 * H/4][4], lazily
    std::vector<unsigned> h_counts;     // per-band candidate counts
    // contours of a finished batch whose results did not fit the caller's
    // buffer: handed out by the next hk_blob_finish*() call
    Contours held;
    bool has_held = false;
};

extern "C" int hk_blob_params_hawkeye(hk_blob_params *p, int scan_height) {
    if (!p) return hk::fail(HK_ERR_ARG, "hk_blob_params_hawkeye: null");
    p->thresholdStep = 10;
    p->minThreshold = 50;
    p->maxThreshold = 220;
    p->minRepeatability = 2;
    p->minDistBetweenBlobs = 10;
    p->filterByColor = 1;
    p->blobColor = 255;
    p->filterByArea = 1;
    p->minArea = scan_height >= 480 ? 7.0f : 1.0f;
    p->maxArea = scan_height >= 480 ? 1000.0f : 150.0f;
    p->filterByCircularity = 1;
    p->minCircularity = 0.3f;
    p->maxCircularity = 1.1f;
    p->filterByInertia = 0;
    p->filterByConvexity = 0;
    return HK_OK;
}

extern "C" int hk_blob_destroy(hk_blob *b) {
    if (!b) return HK_OK;
    hk::DeviceGuard guard(b->device);
    (void)hipFree(b->cands);
    (void)hipFree(b->ctr_set[0]);
    (void)hipFree(b->ctr_set[1]);
    (void)hipFree(b->recs);
    (void)hipFree(b->pts);
    (void)hipFree(b->tmp_pts);
    (void)hipFree(b->active);
    (void)hipFree(b->bits);
    (void)hipFree(b->tiles);
    (void)hipHostFree(b->h_ctr);
    (void)hipHostFree(b->h_recs);
    if (b->done) (void)hipEventDestroy(b->done);
    delete b;
    return HK_OK;
}

extern "C" int hk_blob_create(int width, int height, int max_frames,
                              const hk_blob_params *params, int device,
                              hk_blob **out) {
    if (!out || !params || width <= 0 || height <= 0 || max_frames <= 0)
        return hk::fail(HK_ERR_ARG, "hk_blob_create: bad argument");
    if (width > 65535 || height > 65535)
        return hk::fail(HK_ERR_ARG, "hk_blob_create: frame larger than 65535");
    if (params->filterByInertia || params->filterByConvexity)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_create: inertia/convexity filters are not "
                        "supported (Hawkeye turns them off)");
    if (!params->filterByArea || !params->filterByCircularity ||
        !(params->minCircularity > 0.f) || !(params->maxArea > 0.f))
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_create: filterByArea and filterByCircularity "
                        "(minCircularity > 0) are required: they bound the "
                        "contour length the device trace follows");
    if (params->filterByColor && params->blobColor != 255 &&
        params->blobColor != 0)
        return hk::fail(HK_ERR_ARG, "hk_blob_create: blobColor must be 0/255");
    if (!(params->thresholdStep > 0.f) || params->minRepeatability <= 0)
        return hk::fail(HK_ERR_ARG, "hk_blob_create: bad threshold settings");
    // A contour that passes both filters has perimeter^2 <= 4 pi maxArea /
    // minCircularity and at most `perimeter` points.
    const double max_perim = std::sqrt(4 * 3.14159265358979323846 *
                                       (double)params->maxArea /
                                       (double)params->minCircularity);
    const int max_pts = (int)std::floor(max_perim) + 3;
    // (the tracer's LDS window reaches 32 * COL_MARGIN pixels beyond its
    // block of columns, and a border strays max_pts / 2 from its start)
    constexpr int MAX_PTS_OK =
        MAXPTS < 2 * (32 * COL_MARGIN - 2) ? MAXPTS : 2 * (32 * COL_MARGIN - 2);
    if (max_pts > MAX_PTS_OK)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_create: maxArea/minCircularity allow "
                        "contours of %d points, more than the %d supported",
                        max_pts, MAX_PTS_OK);
    hk::DeviceGuard guard(device);
    if (!guard.ok)
        return hk::fail(HK_ERR_HIP, "hk_blob_create: cannot select device %d",
                        device);
    hk_blob *b = new hk_blob();
    b->width = width;
    b->height = height;
    b->max_frames = max_frames;
    b->device = device;
    b->params = *params;
    b->dev.filterByColor = params->filterByColor;
    b->dev.blobColor = params->blobColor;
    b->dev.minArea = params->minArea;
    b->dev.maxArea = params->maxArea;
    b->dev.minCirc = params->minCircularity;
    b->dev.maxCirc = params->maxCircularity;
    b->dev.max_pts = max_pts;
    for (double t = params->minThreshold; t < params->maxThreshold;
         t += params->thresholdStep)
        ++b->n_thresholds;
    // K3 holds band_rows + max_pts/2 + 4 rows of a WIN_WORDS-wide window of
    // the 1-bit image in LDS at most (14 KB for Hawkeye's parameters; fewer
    // rows when the candidates of a workgroup start close together)
    b->words_per_row = ((width + 31) / 32 + 2 + 3) / 4 * 4;
    {
        const int extra = max_pts / 2 + 4;
        b->band_rows = std::min(64, height);
        b->nbands = (height + b->band_rows - 1) / b->band_rows;
        b->ncols = ((width + 31) / 32 + COL_WORDS - 1) / COL_WORDS;
        // (+1 word: the 64-bit read of a row's last word looks one ahead)
        b->trace_lds = (size_t)4 * (WIN_WORDS * (b->band_rows + extra) + 4);
    }
    const size_t nbuckets = (size_t)max_frames * b->nbands * b->ncols;
    b->cap_blob = (unsigned)max_frames * 128u;

    hipError_t e = hipMalloc((void **)&b->cands,
                             sizeof(Cand) * b->band_cap * nbuckets);
    b->ctr_words = CTR_WORDS + nbuckets;
    for (int k = 0; k < 2; ++k)     // counters, then a candidate count per band
        if (e == hipSuccess)
            e = hipMalloc((void **)&b->ctr_set[k],
                          sizeof(unsigned) * b->ctr_words);
    b->ctr = b->ctr_set[0];
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)blob_trace_k,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                TRACE_LDS_BYTES);
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->recs, sizeof(BlobRec) * b->cap_blob);
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->pts,
                      sizeof(ushort2) * MAXPTS * (size_t)b->cap_blob);
    // a first-pass point list per candidate traced: 256 per frame is a few
    // times what noisy 1080p frames produce after pruning; when they run out
    // a border that passes the filters is followed a second time instead
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->active, sizeof(unsigned) * nbuckets);
    b->cap_tmp = (unsigned)max_frames * 256u;
    if (e == hipSuccess)
        e = hipMalloc((void **)&b->tmp_pts,
                      sizeof(uint32_t) * MAXPTS * (size_t)b->cap_tmp);
    const size_t bit_bytes = sizeof(uint32_t) * (size_t)max_frames *
                             (height + 2) * b->words_per_row;
    if (e == hipSuccess) e = hipMalloc((void **)&b->bits, bit_bytes);
    if (e == hipSuccess) e = hipMemset(b->bits, 0, bit_bytes);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&b->h_ctr, sizeof(unsigned) * CTR_WORDS);
    if (e == hipSuccess)
        e = hipHostGetDevicePointer((void **)&b->h_ctr_dev, b->h_ctr, 0);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&b->h_recs, sizeof(HostRec) * b->cap_blob);
    if (e == hipSuccess)
        e = hipHostGetDevicePointer((void **)&b->hrecs, b->h_recs, 0);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&b->done, hipEventDisableTiming |
                                                  hipEventBlockingSync);
    if (e != hipSuccess) {
        const char *msg = hipGetErrorString(e);
        hk_blob_destroy(b);
        return hk::fail(HK_ERR_ALLOC, "hk_blob_create: %s", msg);
    }
    *out = b;
    return HK_OK;
}

// The kernels of the batch b->masks / nframes on b->stream (device current).
static int launch_batch(hk_blob *b, int nframes) {
    const hipStream_t s = b->stream;
    const uint8_t *masks = b->masks;
    const long frame_stride = b->frame_stride, pitch = b->pitch;

    const unsigned nbuckets = (unsigned)nframes * b->nbands * b->ncols;
    // (the previous batch on this handle was finished before this one was
    // enqueued, whatever stream it ran on)
    const int at = 1 - b->ctr_at;
    b->ctr_at = at;
    b->ctr = b->ctr_set[at];
    if (!b->ctr_clean[at])
        HK_HIP(hipMemsetAsync(b->ctr, 0, sizeof(unsigned) * b->ctr_words, s));
    b->ctr_clean[at] = false;
    b->ctr_clean[1 - at] = false;
    if (!masks) {               // the MOG kernel's 1-bit tiles
        const int tiles_x = b->width / 32, tiles_y = b->height / 4;
        const long total = (long)nframes * tiles_x * tiles_y;
        // few, long-lived workgroups: this runs beside the MOG launch of the
        // next chunk, which leaves wave slots free only now and then
        static const long scan_grid = [] {
            const char *e = getenv("HK_BLOB_SCAN_GRID");
            const int v = e ? atoi(e) : 0;
            return v > 0 ? (long)v : 2048L;
        }();
        const unsigned blocks =
            (unsigned)std::min<long>((total + 255) / 256, scan_grid);
        hipLaunchKernelGGL(blob_scan_tiles_k, dim3(blocks), dim3(256), 0, s,
                           (const uint4 *)b->tiles, tiles_x, tiles_y, nframes,
                           b->height, b->bits, b->words_per_row, b->cands,
                           b->ctr, b->band_rows, b->nbands, b->band_cap,
                           b->ncols, b->active);
        HK_LAUNCH_CHECK("blob_scan_tiles_k");
    } else {
        const int segs = (b->width + 15) / 16;
        const bool vec = ((uintptr_t)masks % 16 == 0) && pitch % 16 == 0 &&
                         (nframes == 1 || frame_stride % 16 == 0);
        const long total = (long)nframes * b->height * segs;
        hipLaunchKernelGGL(blob_scan_k, dim3((unsigned)((total + 255) / 256)),
                           dim3(256), 0, s, masks, frame_stride, pitch,
                           b->width, b->height, nframes, segs, vec, b->cands,
                           b->ctr, b->band_rows, b->nbands,
                           (unsigned short *)b->bits, b->words_per_row,
                           b->band_cap, b->ncols, b->active);
        HK_LAUNCH_CHECK("blob_scan_k");
    }
    // The grid only bounds how many buckets are traced at a time: two
    // workgroups per CU of an MI355X.  Measured beside the MOG launches of
    // the next chunk (detection of the 1080p benchmark clip, ms per clip):
    // 256: 7.95, 384: 7.71, 512: 7.63, 640: 7.78, 1024: 8.07, 2048: 8.50,
    // one workgroup per bucket as before: 8.75.
    static const unsigned trace_grid = [] {
        const char *e = getenv("HK_BLOB_TRACE_GRID");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? (unsigned)v : 512u;
    }();
    hipLaunchKernelGGL(blob_trace_k,
                       dim3(nbuckets < trace_grid ? nbuckets : trace_grid),
                       dim3(TRACE_THREADS), b->trace_lds,
                       s, b->bits, b->words_per_row, b->width, b->height,
                       b->band_rows, b->nbands, b->cands, b->ctr, b->recs,
                       b->pts, b->cap_blob, b->dev, b->band_cap, b->ncols,
                       b->tmp_pts, b->cap_tmp, b->active);
    HK_LAUNCH_CHECK("blob_trace_k");
    static const unsigned radius_grid = [] {
        const char *e = getenv("HK_BLOB_RADIUS_GRID");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? (unsigned)v : 1024u;
    }();
    hipLaunchKernelGGL(blob_radius_k, dim3(radius_grid), dim3(64), 0, s, b->ctr,
                       b->cap_blob, b->recs, b->pts,
                       (const ushort2 *)b->tmp_pts, b->hrecs, b->h_ctr_dev,
                       b->ctr_set[1 - at], (unsigned)b->ctr_words);
    HK_LAUNCH_CHECK("blob_radius_k");
    b->ctr_clean[1 - at] = true;
    HK_HIP(hipEventRecord(b->done, s));
    return HK_OK;
}

extern "C" int hk_blob_enqueue(hk_blob *b, const void *mask_dev,
                               long frame_stride, long pitch, int nframes,
                               void *stream) {
    if (!b || !mask_dev)
        return hk::fail(HK_ERR_ARG, "hk_blob_enqueue: null argument");
    if (nframes <= 0 || nframes > b->max_frames)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_enqueue: nframes %d outside 1..%d", nframes,
                        b->max_frames);
    if (pitch < b->width ||
        (nframes > 1 && frame_stride < pitch * (long)b->height))
        return hk::fail(HK_ERR_ARG, "hk_blob_enqueue: bad pitch/stride");
    if (b->pending_frames)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_enqueue: previous batch not finished");
    hk::DeviceGuard guard(b->device);
    b->has_held = false;
    b->held = Contours();
    b->stream = (hipStream_t)stream;
    b->masks = (const uint8_t *)mask_dev;
    b->frame_stride = frame_stride;
    b->pitch = pitch;
    const int rc = launch_batch(b, nframes);
    if (rc != HK_OK) return rc;
    b->pending_frames = nframes;
    return HK_OK;
}

extern "C" int hk_blob_tiles_buffer(hk_blob *b, void **tiles_dev,
                                    long *words_per_frame) {
    if (!b || !tiles_dev || !words_per_frame)
        return hk::fail(HK_ERR_ARG, "hk_blob_tiles_buffer: null argument");
    if (b->width % 32 || b->height % 4)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_tiles_buffer: %dx%d frames are not made of "
                        "32x4 tiles", b->width, b->height);
    const long wpf = (long)(b->width / 32) * (b->height / 4) * 4;
    if (!b->tiles) {
        hk::DeviceGuard guard(b->device);
        HK_HIP(hipMalloc((void **)&b->tiles,
                         sizeof(uint32_t) * wpf * b->max_frames));
    }
    *tiles_dev = b->tiles;
    *words_per_frame = wpf;
    return HK_OK;
}

extern "C" int hk_blob_enqueue_tiles(hk_blob *b, int nframes, void *stream) {
    if (!b) return hk::fail(HK_ERR_ARG, "hk_blob_enqueue_tiles: null handle");
    if (!b->tiles)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_enqueue_tiles: no tile buffer "
                        "(hk_blob_tiles_buffer)");
    if (nframes <= 0 || nframes > b->max_frames)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_enqueue_tiles: nframes %d outside 1..%d",
                        nframes, b->max_frames);
    if (b->pending_frames)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob_enqueue_tiles: previous batch not finished");
    hk::DeviceGuard guard(b->device);
    b->has_held = false;
    b->held = Contours();
    b->stream = (hipStream_t)stream;
    b->masks = nullptr;
    b->frame_stride = b->pitch = 0;
    const int rc = launch_batch(b, nframes);
    if (rc != HK_OK) return rc;
    b->pending_frames = nframes;
    return HK_OK;
}

static int wait_done(hk_blob *b);

// Grows the candidate lists (to the largest per-band count of the batch) and
// the contour buffers (4x: how many contours pass the filters is only known
// once every candidate has been traced), then runs the batch again.
static int grow_and_rerun(hk_blob *b, int nframes, bool band_full) {
    const size_t nbuckets_max =
        (size_t)b->max_frames * b->nbands * b->ncols;
    HK_HIP(hipStreamSynchronize(b->stream));
    if (band_full) {
        const size_t nb = (size_t)nframes * b->nbands * b->ncols;
        b->h_counts.resize(nb);
        HK_HIP(hipMemcpy(b->h_counts.data(), b->ctr + CTR_WORDS,
                         sizeof(unsigned) * nb, hipMemcpyDeviceToHost));
        unsigned need = 0;
        for (unsigned v : b->h_counts) need = std::max(need, v);
        unsigned cap = b->band_cap;
        while (cap < need) cap *= 2;
        (void)hipFree(b->cands);
        b->cands = nullptr;
        b->band_cap = cap;
        HK_HIP(hipMalloc((void **)&b->cands,
                         sizeof(Cand) * (size_t)cap * nbuckets_max));
    }
    if (b->h_ctr[CTR_NBLOB] > b->cap_blob) {
        unsigned cap = b->cap_blob;
        while (cap < b->h_ctr[CTR_NBLOB]) cap *= 4;
        (void)hipFree(b->recs);
        (void)hipFree(b->pts);
        (void)hipHostFree(b->h_recs);
        b->recs = nullptr;
        b->hrecs = nullptr;
        b->pts = nullptr;
        b->h_recs = nullptr;
        b->cap_blob = cap;
        HK_HIP(hipMalloc((void **)&b->recs, sizeof(BlobRec) * cap));
        HK_HIP(hipMalloc((void **)&b->pts,
                         sizeof(ushort2) * MAXPTS * (size_t)cap));
        HK_HIP(hipHostMalloc((void **)&b->h_recs, sizeof(HostRec) * cap));
        HK_HIP(hipHostGetDevicePointer((void **)&b->hrecs, b->h_recs, 0));
    }
    const int rc = launch_batch(b, nframes);
    if (rc != HK_OK) return rc;
    return wait_done(b);
}

// Waits for the batch's `done` event, but not for ever: a batch is a few
// hundred microseconds of GPU work behind at most a few MOG launches, so an
// event that has not fired after HK_BLOB_WAIT_S seconds (default 60) means
// the work in front of it is stuck (a stream dependency that cannot resolve,
// a faulted queue).  The caller then gets HK_ERR_HIP instead of a thread that
// never returns.
static int wait_done(hk_blob *b) {
    static const double limit_s = [] {
        const char *e = getenv("HK_BLOB_WAIT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 60.0;
    }();
    static const double spin_s = [] {
        const char *e = getenv("HK_BLOB_SPIN_US");
        return e ? atof(e) * 1e-6 : 0.0;
    }();
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    for (long spins = 0;; ++spins) {
        const hipError_t e = hipEventQuery(b->done);
        if (e == hipSuccess) return HK_OK;
        if (e != hipErrorNotReady)
            return hk::fail(HK_ERR_HIP, "hk_blob: hipEventQuery: %s",
                            hipGetErrorString(e));
        if (spins < 200) continue;              // ~ the first 100 us: spin
        const double waited =
            std::chrono::duration<double>(clock::now() - t0).count();
        if (waited < spin_s) continue;
        if (waited > limit_s)
            return hk::fail(HK_ERR_HIP,
                            "hk_blob: the batch did not finish within %.0f s "
                            "(stream %p); GPU work in front of it is stuck",
                            limit_s, (void *)b->stream);
        // a batch normally ends within a MOG launch (< 1 ms): stay on the
        // core for a few of those, then get out of the way
        if (waited < 0.003) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(
                 waited < 0.1 ? 50 : 1000));
    }
}

// Waits for the batch and returns the surviving contours per frame in the
// order findContours(RETR_LIST) lists them (reverse discovery order).
static int collect(hk_blob *b, Contours &frames) {
    if (!b) return hk::fail(HK_ERR_ARG, "hk_blob_finish: null handle");
    if (!b->pending_frames && b->has_held) {
        frames.swap(b->held);
        b->has_held = false;
        return HK_OK;
    }
    if (!b->pending_frames)
        return hk::fail(HK_ERR_ARG, "hk_blob_finish: nothing enqueued");
    b->has_held = false;
    hk::DeviceGuard guard(b->device);
    const int nframes = b->pending_frames;
    b->pending_frames = 0;
    {
        const int rc = wait_done(b);
        if (rc != HK_OK) return rc;
    }
    if (b->h_ctr[CTR_FLAGS] & FLAG_NONBINARY)
        return hk::fail(HK_ERR_ARG,
                        "hk_blob: mask holds values other than 0 and 255");
    // A list that overflowed (very noisy masks): the counters hold the sizes
    // that were needed, so the buffers grow to those and the batch runs once
    // more.  The buffers keep their size afterwards.
    for (int pass = 0; pass < 3; ++pass) {
        const bool band_full = (b->h_ctr[CTR_FLAGS] & FLAG_BAND_FULL) != 0;
        const bool blob_full = b->h_ctr[CTR_NBLOB] > b->cap_blob;
        if (!band_full && !blob_full) break;
        if (pass == 2)
            return hk::fail(HK_ERR_INTERNAL,
                            "hk_blob: work lists still full after growing");
        const int rc = grow_and_rerun(b, nframes, band_full);
        if (rc != HK_OK) return rc;
    }
    const unsigned nblob = b->h_ctr[CTR_NBLOB];
    // The records the radius kernel left for the host (32 bytes each, pinned
    // memory): by frame (counting sort), within a frame last discovered
    // first (descending key; a few dozen per frame).
    const HostRec *const lite = b->h_recs;
    static thread_local std::vector<unsigned> slot, fill;
    std::vector<int> &off = frames.off;
    off.assign((size_t)nframes + 1, 0);
    for (unsigned i = 0; i < nblob; ++i) {
        const int f = lite[i].frame;
        if (f < 0 || f >= nframes)
            return hk::fail(HK_ERR_INTERNAL, "hk_blob: bad frame index");
        ++off[(size_t)f + 1];
    }
    for (int f = 0; f < nframes; ++f) off[(size_t)f + 1] += off[(size_t)f];
    slot.resize(nblob);
    fill.assign(off.begin(), off.end() - 1);
    for (unsigned i = 0; i < nblob; ++i)
        slot[fill[(size_t)lite[i].frame]++] = i;
    frames.centers.resize(nblob);
    for (int f = 0; f < nframes; ++f) {
        unsigned *lo = slot.data() + off[(size_t)f];
        unsigned *hi = slot.data() + off[(size_t)f + 1];
        for (unsigned *p = lo + 1; p < hi; ++p) {       // insertion sort
            const unsigned v = *p;
            const uint32_t key = lite[v].key;
            unsigned *q = p;
            while (q > lo && lite[q[-1]].key < key) {
                *q = q[-1];
                --q;
            }
            *q = v;
        }
        for (unsigned *p = lo; p < hi; ++p) {
            const HostRec &r = lite[*p];
            Center &c = frames.centers[(size_t)(p - slot.data())];
            c.x = r.cx;
            c.y = r.cy;
            c.radius = r.radius;
        }
    }
    return HK_OK;
}

extern "C" int hk_blob_finish_centers(hk_blob *b, int *counts_host,
                                      double *centers_host, int cap,
                                      int *total) {
    if (!counts_host || !centers_host || !total)
        return hk::fail(HK_ERR_ARG, "hk_blob_finish_centers: null argument");
    Contours frames;
    const int rc = collect(b, frames);
    if (rc != HK_OK) return rc;
    const int n = (int)frames.centers.size();
    for (size_t f = 0; f < frames.frames(); ++f)
        counts_host[f] = frames.off[f + 1] - frames.off[f];
    for (int i = 0; i < n && i < cap; ++i) {
        const Center &c = frames.centers[(size_t)i];
        centers_host[3 * i] = c.x;
        centers_host[3 * i + 1] = c.y;
        centers_host[3 * i + 2] = c.radius;
    }
    *total = n;
    if (n > cap) {
        b->held.swap(frames);
        b->has_held = true;
        return hk::fail(HK_ERR_CAPACITY, "hk_blob: %d centres, capacity %d",
                        n, cap);
    }
    return HK_OK;
}

extern "C" int hk_blob_finish(hk_blob *b, int *counts_host, float *kp_host,
                              int cap, int *total) {
    if (!counts_host || !kp_host || !total)
        return hk::fail(HK_ERR_ARG, "hk_blob_finish: null argument");
    Contours frames;
    const int rc = collect(b, frames);
    if (rc != HK_OK) return rc;
    int n = 0;
    static thread_local std::vector<float> kp;
    for (size_t f = 0; f < frames.frames(); ++f) {
        kp.clear();
        group_keypoints(frames.centers.data() + frames.off[f],
                        (size_t)(frames.off[f + 1] - frames.off[f]),
                        b->n_thresholds, b->params, kp);
        const int k = (int)(kp.size() / 3);
        counts_host[f] = k;
        for (int i = 0; i < k; ++i, ++n)
            if (n < cap) std::memcpy(kp_host + 3 * (size_t)n, &kp[3 * i],
                                     3 * sizeof(float));
    }
    *total = n;
    if (n > cap) {
        b->held.swap(frames);
        b->has_held = true;
        return hk::fail(HK_ERR_CAPACITY, "hk_blob: %d keypoints, capacity %d",
                        n, cap);
    }
    return HK_OK;
}

extern "C" int hk_blob_detect(hk_blob *b, const void *mask_dev,
                              long frame_stride, long pitch, int nframes,
                              void *stream, int *counts_host, float *kp_host,
                              int cap, int *total) {
    const int rc = hk_blob_enqueue(b, mask_dev, frame_stride, pitch, nframes,
                                   stream);
    if (rc != HK_OK) return rc;
    return hk_blob_finish(b, counts_host, kp_host, cap, total);
}

