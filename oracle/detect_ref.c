/*
 * ORACLE -- test infrastructure only.  Never linked into or called by the
 * product (hawkeye_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load this library.
 *
 * CPU restatement of the detection half of the reference's step 2:
 *
 *   fgbg = cv2.bgsegm.createBackgroundSubtractorMOG()   reference tracker.py:357
 *   fgmask = fgbg.apply(frame)                          reference tracker.py:395
 *   detector = cv2.SimpleBlobDetector_create(params)    reference tracker.py:359-374
 *   keypoints = detector.detect(fgmask)                 reference tracker.py:396
 *
 * The arithmetic of those four calls lives in a third-party dependency that is
 * NOT under /root/reference: OpenCV 3.4.4 + opencv_contrib (bgsegm), pinned by
 * the reference's environment_osx.yml:44 / environment_win.yml:26.  No OpenCV
 * source, header or binary exists in the build image, and the reference has no
 * tests or golden vectors at this boundary, so
 *
 *                     *** PARITY UNPINNED ***
 *
 * for this file: it restates the published algorithms of
 *   opencv_contrib/modules/bgsegm/src/bgfg_gaussmix.cpp  (process8uC3, apply)
 *   opencv/modules/imgproc/src/contours.cpp   (cvFindNextContour, icvFetchContour,
 *                                              RETR_LIST + CHAIN_APPROX_NONE)
 *   opencv/modules/imgproc/src/moments.cpp    (contourMoments)
 *   opencv/modules/imgproc/src/shapedescr.cpp (arcLength)
 *   opencv/modules/features2d/src/blobdetector.cpp (findBlobs, detect)
 * from knowledge of those files, operation by operation and in the same
 * order (all float/double expressions are written so that a compiler without
 * FMA contraction evaluates them exactly as the x86-64 SSE2 build of OpenCV
 * does; build with -ffp-contract=off).  tests/test_oracle_detect.py checks it
 * against hand-computable cases and against an independent topological
 * count of borders; tests against a real cv2 auto-skip when cv2.bgsegm is not
 * importable.
 */

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/*  MOG (KaewTraKulPong & Bowden mixture), 8UC3 path                        */
/* ------------------------------------------------------------------------ */

/* std::max(a, b) as OpenCV writes it */
#define MAXF(a, b) ((a) < (b) ? (b) : (a))

#define MOG_K 5                 /* defaultNMixtures */
#define MOG_HISTORY 200         /* defaultHistory */

typedef struct {
    float sortKey;
    float weight;
    float mean[3];
    float var[3];
} MixData;

typedef struct {
    int width, height;
    int nframes;
    MixData *model;             /* [height*width][MOG_K], zero initialised */
} MogRef;

void *hkref_mog_create(int width, int height)
{
    MogRef *m = (MogRef *)calloc(1, sizeof(MogRef));
    if (!m)
        return NULL;
    m->width = width;
    m->height = height;
    m->model = (MixData *)calloc((size_t)width * height * MOG_K,
                                 sizeof(MixData));
    if (!m->model) {
        free(m);
        return NULL;
    }
    return m;
}

void hkref_mog_destroy(void *h)
{
    MogRef *m = (MogRef *)h;
    if (m) {
        free(m->model);
        free(m);
    }
}

int hkref_mog_nframes(void *h) { return ((MogRef *)h)->nframes; }
const float *hkref_mog_model(void *h)
{
    return (const float *)((MogRef *)h)->model;
}

/* Test hook: continue from a given state (model in MixData order,
 * [height*width][MOG_K][8] floats, and the number of frames it has seen), so
 * that a window deep inside a long clip can be checked without the oracle
 * running the thousand frames before it. */
void hkref_mog_set_state(void *h, const float *model, int nframes)
{
    MogRef *m = (MogRef *)h;
    memcpy(m->model, model,
           (size_t)m->width * m->height * MOG_K * sizeof(MixData));
    m->nframes = nframes;
}

/* Learning rate used for frame number n (1-based) when the caller passes
 * learningRate < 0, which is what the Python binding's apply(frame) does. */
float hkref_mog_alpha(int nframes)
{
    int d = nframes < MOG_HISTORY ? nframes : MOG_HISTORY;
    return (float)(1. / d);
}

void hkref_mog_apply(void *h, const uint8_t *bgr, long pitch, uint8_t *mask,
                     long mask_pitch)
{
    MogRef *m = (MogRef *)h;
    const int K = MOG_K;
    const float alpha = hkref_mog_alpha(++m->nframes);
    const float T = (float)0.7;            /* defaultBackgroundRatio */
    const float vT = (float)(2.5 * 2.5);   /* defaultVarThreshold */
    const float w0 = (float)0.05;          /* defaultInitialWeight */
    const float sk0 = (float)(w0 / (15. * 2 * sqrt(3.)));
    const float var0 = (float)(15. * 15. * 4);
    const float minVar = (float)(15. * 15.);   /* noiseSigma 0 -> 15 */
    MixData *mptr = m->model;

    for (int y = 0; y < m->height; y++) {
        const uint8_t *src = bgr + (size_t)y * pitch;
        uint8_t *dst = mask + (size_t)y * mask_pitch;
        for (int x = 0; x < m->width; x++, mptr += K) {
            float wsum = 0;
            float pix[3] = { (float)src[x * 3], (float)src[x * 3 + 1],
                             (float)src[x * 3 + 2] };
            int kHit = -1, kForeground = -1;
            int k, k1;

            for (k = 0; k < K; k++) {
                float w = mptr[k].weight;
                wsum += w;
                if (w < FLT_EPSILON)
                    break;
                float d0 = pix[0] - mptr[k].mean[0];
                float d1 = pix[1] - mptr[k].mean[1];
                float d2_ = pix[2] - mptr[k].mean[2];
                float v0 = mptr[k].var[0], v1 = mptr[k].var[1],
                      v2 = mptr[k].var[2];
                float dist2 = 0;
                dist2 += d0 * d0;
                dist2 += d1 * d1;
                dist2 += d2_ * d2_;
                if (dist2 < vT * (v0 + v1 + v2)) {
                    wsum -= w;
                    float dw = alpha * (1.f - w);
                    mptr[k].weight = w + dw;
                    mptr[k].mean[0] = mptr[k].mean[0] + alpha * d0;
                    mptr[k].mean[1] = mptr[k].mean[1] + alpha * d1;
                    mptr[k].mean[2] = mptr[k].mean[2] + alpha * d2_;
                    v0 = MAXF(v0 + alpha * (d0 * d0 - v0), minVar);
                    v1 = MAXF(v1 + alpha * (d1 * d1 - v1), minVar);
                    v2 = MAXF(v2 + alpha * (d2_ * d2_ - v2), minVar);
                    mptr[k].var[0] = v0;
                    mptr[k].var[1] = v1;
                    mptr[k].var[2] = v2;
                    mptr[k].sortKey = w / sqrtf(v0 + v1 + v2);

                    for (k1 = k - 1; k1 >= 0; k1--) {
                        if (mptr[k1].sortKey >= mptr[k1 + 1].sortKey)
                            break;
                        MixData t = mptr[k1];
                        mptr[k1] = mptr[k1 + 1];
                        mptr[k1 + 1] = t;
                    }
                    kHit = k1 + 1;
                    break;
                }
            }

            if (kHit < 0) {
                /* nothing matched: replace the weakest (or first empty) */
                kHit = k = k < K - 1 ? k : K - 1;
                wsum += w0 - mptr[k].weight;
                mptr[k].weight = w0;
                mptr[k].mean[0] = pix[0];
                mptr[k].mean[1] = pix[1];
                mptr[k].mean[2] = pix[2];
                mptr[k].var[0] = mptr[k].var[1] = mptr[k].var[2] = var0;
                mptr[k].sortKey = sk0;
            } else {
                /* k is the pre-reorder index of the hit; the tail is read
                 * from the already reordered array, exactly as upstream */
                for (; k < K; k++)
                    wsum += mptr[k].weight;
            }

            float wscale = 1.f / wsum;
            wsum = 0;
            for (k = 0; k < K; k++) {
                mptr[k].weight *= wscale;
                wsum += mptr[k].weight;
                mptr[k].sortKey *= wscale;
                if (wsum > T && kForeground < 0)
                    kForeground = k + 1;
            }
            dst[x] = (uint8_t)(-(kHit >= kForeground));
        }
    }
}

/* ------------------------------------------------------------------------ */
/*  findContours(RETR_LIST, CHAIN_APPROX_NONE)                              */
/* ------------------------------------------------------------------------ */

typedef struct {
    int *pts;        /* x0,y0,x1,y1,... for all contours back to back */
    long npts, cap_pts;
    long *start;     /* start[i] = first point index of contour i */
    int *is_hole;
    int n, cap_n;
} ContourList;

static void cl_init(ContourList *c)
{
    memset(c, 0, sizeof(*c));
}

static void cl_free(ContourList *c)
{
    free(c->pts);
    free(c->start);
    free(c->is_hole);
    memset(c, 0, sizeof(*c));
}

static void cl_begin(ContourList *c, int is_hole)
{
    if (c->n + 2 > c->cap_n) {
        c->cap_n = c->cap_n ? c->cap_n * 2 : 64;
        c->start = (long *)realloc(c->start, sizeof(long) * (c->cap_n + 1));
        c->is_hole = (int *)realloc(c->is_hole, sizeof(int) * c->cap_n);
    }
    c->start[c->n] = c->npts;
    c->is_hole[c->n] = is_hole;
    c->n++;
    c->start[c->n] = c->npts;
}

static void cl_point(ContourList *c, int x, int y)
{
    if (c->npts + 1 > c->cap_pts) {
        c->cap_pts = c->cap_pts ? c->cap_pts * 2 : 1024;
        c->pts = (int *)realloc(c->pts, sizeof(int) * 2 * c->cap_pts);
    }
    c->pts[2 * c->npts] = x;
    c->pts[2 * c->npts + 1] = y;
    c->npts++;
    c->start[c->n] = c->npts;
}

/* Border following from i0 (padded-image pointer) at padded coordinates
 * (px,py); emitted points are shifted by (-1,-1) like cv::findContours does. */
static void fetch_contour(signed char *i0, int step, int px, int py,
                          int is_hole, ContourList *out)
{
    static const int dx8[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
    static const int dy8[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };
    const signed char nbd = 2;
    int deltas[16];
    for (int i = 0; i < 8; i++)
        deltas[i] = deltas[i + 8] = dy8[i] * step + dx8[i];

    signed char *i1, *i3, *i4 = 0;
    int s, s_end;
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);

    if (s == s_end) {                       /* single pixel domain */
        *i0 = (signed char)(nbd | -128);
        cl_point(out, px - 1, py - 1);
        return;
    }

    i3 = i0;
    for (;;) {
        s_end = s;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0)
                break;
        }
        s &= 7;

        if ((unsigned)(s - 1) < (unsigned)s_end)    /* "right" bound */
            *i3 = (signed char)(nbd | -128);
        else if (*i3 == 1)
            *i3 = nbd;

        cl_point(out, px - 1, py - 1);      /* CHAIN_APPROX_NONE */
        px += dx8[s];
        py += dy8[s];

        if (i4 == i0 && i3 == i1)
            break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* Contours in DISCOVERY order into `out`; the public wrapper reverses. */
static int find_contours_raw(const uint8_t *img, int w, int h, long pitch,
                             ContourList *out)
{
    const int pw = w + 2, ph = h + 2;
    signed char *pad = (signed char *)calloc((size_t)pw * ph, 1);
    if (!pad)
        return -1;
    for (int y = 0; y < h; y++) {
        const uint8_t *src = img + (size_t)y * pitch;
        signed char *dst = pad + (size_t)(y + 1) * pw + 1;
        for (int x = 0; x < w; x++)
            dst[x] = src[x] ? 1 : 0;
    }

    const int width = pw - 1, height = ph - 1;     /* scanner->img_size */
    for (int y = 1; y < height; y++) {
        signed char *row = pad + (size_t)y * pw;
        int prev = 0;
        int x = 1;
        for (; x < width; x++) {
            int p = 0;
            for (; x < width && (p = row[x]) == prev; x++)
                ;
            if (x >= width)
                break;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {           /* not an outer border */
                if (p != 0 || prev < 1)             /* nor a hole border */
                    goto resume_scan;
                is_hole = 1;
            }
            cl_begin(out, is_hole);
            fetch_contour(row + x - is_hole, pw, x - is_hole, y, is_hole, out);
            p = row[x];     /* scan resumes at x+1 with prev = img[x] */
resume_scan:
            prev = p;
        }
    }
    free(pad);
    return 0;
}

/* Public: contours in the order cv::findContours returns them for RETR_LIST
 * (reverse discovery order).  Caller frees with hkref_free(). */
int hkref_find_contours(const uint8_t *img, int w, int h, long pitch,
                        int **pts_out, long **start_out, int **hole_out,
                        int *n_out)
{
    ContourList cl;
    cl_init(&cl);
    if (find_contours_raw(img, w, h, pitch, &cl) != 0)
        return -1;
    int n = cl.n;
    int *pts = (int *)malloc(sizeof(int) * 2 * (cl.npts ? cl.npts : 1));
    long *start = (long *)malloc(sizeof(long) * (n + 1));
    int *hole = (int *)malloc(sizeof(int) * (n ? n : 1));
    long pos = 0;
    for (int i = 0; i < n; i++) {
        int j = n - 1 - i;
        long cnt = cl.start[j + 1] - cl.start[j];
        start[i] = pos;
        hole[i] = cl.is_hole[j];
        memcpy(pts + 2 * pos, cl.pts + 2 * cl.start[j], sizeof(int) * 2 * cnt);
        pos += cnt;
    }
    start[n] = pos;
    cl_free(&cl);
    *pts_out = pts;
    *start_out = start;
    *hole_out = hole;
    *n_out = n;
    return 0;
}

void hkref_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/*  SimpleBlobDetector                                                       */
/* ------------------------------------------------------------------------ */

typedef struct {
    float thresholdStep, minThreshold, maxThreshold;
    int minRepeatability;
    float minDistBetweenBlobs;
    int filterByColor;
    int blobColor;
    int filterByArea;
    float minArea, maxArea;
    int filterByCircularity;
    float minCircularity, maxCircularity;
} HkrefBlobParams;

/* The parameter block Hawkeye builds (reference tracker.py:359-374): library
 * defaults except colour 255, circularity [0.3, 1.1), area by scan height,
 * inertia and convexity filters off. */
void hkref_blob_params_hawkeye(HkrefBlobParams *p, int scan_height)
{
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
}

typedef struct {
    double x, y, radius, confidence;
} Center;

typedef struct {
    Center *v;
    int n, cap;
} CenterVec;

static void cv_push(CenterVec *c, Center e)
{
    if (c->n + 1 > c->cap) {
        c->cap = c->cap ? c->cap * 2 : 8;
        c->v = (Center *)realloc(c->v, sizeof(Center) * c->cap);
    }
    c->v[c->n++] = e;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* One findBlobs() pass on a binary image. */
static int find_blobs(const uint8_t *bin, int w, int h, long pitch,
                      const HkrefBlobParams *params, CenterVec *centers)
{
    int *pts;
    long *start;
    int *hole;
    int n;
    centers->n = 0;
    if (hkref_find_contours(bin, w, h, pitch, &pts, &start, &hole, &n) != 0)
        return -1;

    double *dists = NULL;
    long dcap = 0;
    for (int ci = 0; ci < n; ci++) {
        const int *p = pts + 2 * start[ci];
        const long lpt = start[ci + 1] - start[ci];
        Center center;
        center.confidence = 1;

        /* contourMoments(): Green sums in double */
        double a00 = 0, a10 = 0, a01 = 0;
        double m00 = 0, m10 = 0, m01 = 0;
        if (lpt > 0) {
            double xi_1 = p[2 * (lpt - 1)], yi_1 = p[2 * (lpt - 1) + 1];
            for (long i = 0; i < lpt; i++) {
                double xi = p[2 * i], yi = p[2 * i + 1];
                double dxy = xi_1 * yi - xi * yi_1;
                double xii_1 = xi_1 + xi;
                double yii_1 = yi_1 + yi;
                a00 += dxy;
                a10 += dxy * xii_1;
                a01 += dxy * yii_1;
                xi_1 = xi;
                yi_1 = yi;
            }
            if (fabs(a00) > FLT_EPSILON) {
                double db1_2, db1_6;
                if (a00 > 0) {
                    db1_2 = 0.5;
                    db1_6 = 0.16666666666666666666666666666667;
                } else {
                    db1_2 = -0.5;
                    db1_6 = -0.16666666666666666666666666666667;
                }
                m00 = a00 * db1_2;
                m10 = a10 * db1_6;
                m01 = a01 * db1_6;
            }
        }

        if (params->filterByArea) {
            double area = m00;
            if (area < params->minArea || area >= params->maxArea)
                continue;
        }

        if (params->filterByCircularity) {
            double area = m00;
            /* arcLength(closed): float sqrt per segment, double accumulate */
            double perimeter = 0;
            if (lpt > 1) {
                float prevx = (float)p[2 * (lpt - 1)],
                      prevy = (float)p[2 * (lpt - 1) + 1];
                for (long i = 0; i < lpt; i++) {
                    float px = (float)p[2 * i], py = (float)p[2 * i + 1];
                    float dx = px - prevx, dy = py - prevy;
                    perimeter += sqrtf(dx * dx + dy * dy);
                    prevx = px;
                    prevy = py;
                }
            }
            double ratio = 4 * 3.1415926535897932384626433832795 * area /
                           (perimeter * perimeter);
            if (ratio < params->minCircularity ||
                ratio >= params->maxCircularity)
                continue;
        }

        if (m00 == 0.0)
            continue;
        center.x = m10 / m00;
        center.y = m01 / m00;

        if (params->filterByColor) {
            long ry = lrint(center.y), rx = lrint(center.x);   /* cvRound */
            if (bin[(size_t)ry * pitch + rx] != params->blobColor)
                continue;
        }

        if (lpt > dcap) {
            dcap = lpt * 2;
            dists = (double *)realloc(dists, sizeof(double) * dcap);
        }
        for (long i = 0; i < lpt; i++) {
            double ddx = center.x - (double)p[2 * i];
            double ddy = center.y - (double)p[2 * i + 1];
            dists[i] = sqrt(ddx * ddx + ddy * ddy);
        }
        qsort(dists, lpt, sizeof(double), cmp_double);
        center.radius = (dists[(lpt - 1) / 2] + dists[lpt / 2]) / 2.;
        cv_push(centers, center);
    }
    free(dists);
    free(pts);
    free(start);
    free(hole);
    return 0;
}

/* Centres of ONE findBlobs pass on img thresholded at `thresh` (x,y,radius
 * triples of doubles, OpenCV order).  For tests of the per-contour stage. */
int hkref_find_blobs(const uint8_t *img, int w, int h, long pitch,
                     const HkrefBlobParams *params, double thresh,
                     double *out, int cap)
{
    uint8_t *bin = (uint8_t *)malloc((size_t)w * h);
    int ithresh = (int)floor(thresh);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            bin[(size_t)y * w + x] =
                img[(size_t)y * pitch + x] > ithresh ? 255 : 0;
    CenterVec cur = { 0, 0, 0 };
    int rc = find_blobs(bin, w, h, w, params, &cur);
    free(bin);
    if (rc != 0)
        return -1;
    int n = cur.n;
    for (int i = 0; i < n && i < cap; i++) {
        out[3 * i] = cur.v[i].x;
        out[3 * i + 1] = cur.v[i].y;
        out[3 * i + 2] = cur.v[i].radius;
    }
    free(cur.v);
    return n;
}

/* detect(): all thresholds, grouping across thresholds, keypoints out as
 * float32 (x, y, size) triples.  Returns the keypoint count (may exceed cap,
 * only cap are written) or -1. */
int hkref_blob_detect(const uint8_t *img, int w, int h, long pitch,
                      const HkrefBlobParams *params, float *out, int cap)
{
    CenterVec *groups = NULL;       /* centers: vector<vector<Center>> */
    int ngroups = 0, gcap = 0;
    uint8_t *bin = (uint8_t *)malloc((size_t)w * h);
    CenterVec cur = { 0, 0, 0 };
    if (!bin)
        return -1;

    for (double thresh = params->minThreshold; thresh < params->maxThreshold;
         thresh += params->thresholdStep) {
        int ithresh = (int)floor(thresh);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                bin[(size_t)y * w + x] =
                    img[(size_t)y * pitch + x] > ithresh ? 255 : 0;
        if (find_blobs(bin, w, h, w, params, &cur) != 0) {
            free(bin);
            return -1;
        }

        const int existing = ngroups;   /* new groups join after the pass */
        CenterVec *fresh = NULL;
        int nfresh = 0;
        for (int i = 0; i < cur.n; i++) {
            int isNew = 1;
            for (int j = 0; j < existing; j++) {
                CenterVec *g = &groups[j];
                Center *mid = &g->v[g->n / 2];
                double ddx = mid->x - cur.v[i].x, ddy = mid->y - cur.v[i].y;
                double dist = sqrt(ddx * ddx + ddy * ddy);
                isNew = dist >= params->minDistBetweenBlobs &&
                        dist >= mid->radius && dist >= cur.v[i].radius;
                if (!isNew) {
                    cv_push(g, cur.v[i]);
                    int k = g->n - 1;
                    while (k > 0 && cur.v[i].radius < g->v[k - 1].radius) {
                        g->v[k] = g->v[k - 1];
                        k--;
                    }
                    g->v[k] = cur.v[i];
                    break;
                }
            }
            if (isNew) {
                fresh = (CenterVec *)realloc(fresh,
                                             sizeof(CenterVec) * (nfresh + 1));
                memset(&fresh[nfresh], 0, sizeof(CenterVec));
                cv_push(&fresh[nfresh], cur.v[i]);
                nfresh++;
            }
        }
        if (ngroups + nfresh > gcap) {
            gcap = (ngroups + nfresh) * 2;
            groups = (CenterVec *)realloc(groups, sizeof(CenterVec) * gcap);
        }
        for (int i = 0; i < nfresh; i++)
            groups[ngroups++] = fresh[i];
        free(fresh);
    }

    int nkp = 0;
    for (int i = 0; i < ngroups; i++) {
        CenterVec *g = &groups[i];
        if (g->n >= params->minRepeatability) {
            double sx = 0, sy = 0, normalizer = 0;
            for (int j = 0; j < g->n; j++) {
                sx += g->v[j].confidence * g->v[j].x;
                sy += g->v[j].confidence * g->v[j].y;
                normalizer += g->v[j].confidence;
            }
            double inv = 1. / normalizer;
            sx *= inv;
            sy *= inv;
            if (nkp < cap) {
                out[3 * nkp] = (float)sx;
                out[3 * nkp + 1] = (float)sy;
                out[3 * nkp + 2] = (float)(g->v[g->n / 2].radius) * 2.0f;
            }
            nkp++;
        }
        free(g->v);
    }
    free(groups);
    free(cur.v);
    free(bin);
    return nkp;
}
