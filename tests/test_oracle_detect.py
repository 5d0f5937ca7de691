"""
The detection-half oracle (oracle/detect_ref.c): hand-computable cases, an
independent topological check of the border-following restatement, and -- only
when a real OpenCV with bgsegm is importable -- direct parity with cv2.

OpenCV is not in the build image, so the cv2 tests skip there and the oracle is
"parity unpinned" (see DESIGN.md); everything else runs on CPU in seconds.
"""
import math

import numpy as np
import pytest
from scipy import ndimage

from oracle import detect_ref as D


def square(h, w, y0, x0, n, val=255):
    img = np.zeros((h, w), np.uint8)
    img[y0:y0 + n, x0:x0 + n] = val
    return img


def green_area(pts):
    x, y = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
    xp, yp = np.roll(x, 1), np.roll(y, 1)
    return abs(int(np.sum(xp * y - x * yp))) / 2


# ---------------------------------------------------------------- contours

def test_single_pixel_and_square():
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    cs = D.find_contours(img)
    assert len(cs) == 1 and not cs[0][0]
    assert cs[0][1].tolist() == [[4, 4]]

    for n in (2, 3, 5, 8):
        cs = D.find_contours(square(20, 20, 3, 4, n))
        assert len(cs) == 1
        pts = cs[0][1]
        assert len(pts) == 4 * (n - 1)
        assert green_area(pts) == (n - 1)**2
        assert pts[0].tolist() == [4, 3]            # raster-first pixel


def test_blob_touching_image_edge():
    # cv::findContours pads by one zero pixel, so edge blobs are traced as-is
    cs = D.find_contours(square(6, 6, 0, 0, 3))
    assert len(cs) == 1 and len(cs[0][1]) == 8
    assert green_area(cs[0][1]) == 4


def test_ring_has_hole_contour_and_reverse_order():
    img = square(12, 12, 2, 2, 7)
    img[4:7, 4:7] = 0
    cs = D.find_contours(img)
    assert [c[0] for c in cs] == [True, False]      # last discovered first
    hole = cs[0][1]
    assert hole[0].tolist() == [3, 4]               # (x - 1, y) of first hole px
    # 4-neighbours of the hole only: the four inner corner pixels are cut
    assert len(hole) == 12 and green_area(hole) == 14
    assert green_area(cs[1][1]) == 36


def test_thin_line_is_walked_out_and_back():
    img = np.zeros((5, 12), np.uint8)
    img[2, 2:9] = 255
    (hole, pts), = D.find_contours(img)
    assert len(pts) == 12 and green_area(pts) == 0


@pytest.mark.parametrize('seed,density', [(1, 0.2), (2, 0.45), (3, 0.6),
                                          (4, 0.8), (5, 0.5)])
def test_border_count_matches_topology(seed, density):
    """#contours == #(8-connected foreground components) + #(4-connected
    background components that do not reach the padded frame)."""
    rng = np.random.RandomState(seed)
    img = (rng.uniform(size=(61, 83)) < density).astype(np.uint8) * 255
    if seed == 5:       # smoother blobs
        img = (ndimage.uniform_filter(rng.uniform(size=(61, 83)), 5) >
               0.5).astype(np.uint8) * 255
    cs = D.find_contours(img)
    fg = img > 0
    lab, n_fg = ndimage.label(fg, structure=np.ones((3, 3), int))
    pad = np.pad(~fg, 1, constant_values=True)
    bl, n_bg = ndimage.label(pad)                    # 4-connectivity
    assert len(cs) == n_fg + (n_bg - 1)
    assert sum(1 for c in cs if not c[0]) == n_fg

    # outer borders start at the raster-first pixel of distinct components,
    # discovery order is raster order, list order is its reverse
    firsts = {}
    ys, xs = np.nonzero(fg)
    for y, x in zip(ys, xs):
        firsts.setdefault(lab[y, x], (y, x))
    outer = [c[1] for c in cs if not c[0]]
    starts = [(int(p[0][1]), int(p[0][0])) for p in outer]
    assert sorted(starts) == sorted(firsts.values())
    assert starts == sorted(starts, reverse=True)

    for is_hole, pts in cs:
        assert fg[pts[:, 1], pts[:, 0]].all()
        if len(pts) > 1:
            step = np.abs(pts - np.roll(pts, 1, axis=0)).max(axis=1)
            assert step.max() == 1 and step.min() == 1


# ---------------------------------------------------------------- blobs

def test_square_blob_numbers():
    p = D.hawkeye_params(720)
    img = square(40, 40, 10, 10, 6)                 # polygon area 25
    c = D.find_blobs(img, p)
    assert c.shape == (1, 3)
    assert c[0, 0] == 12.5 and c[0, 1] == 12.5
    # 20 border points: distances sqrt(2.5^2 + k^2), k in {.5, 1.5, 2.5}
    d = sorted([math.hypot(2.5, k) for k in (0.5, 1.5, 2.5)
                for _ in range(8)][:20])
    kp = D.blob_detect(img, p)
    assert kp.shape == (1, 3)
    assert kp[0, 0] == np.float32(12.5) and kp[0, 1] == np.float32(12.5)
    assert kp[0, 2] == np.float32(c[0, 2]) * np.float32(2.0)
    assert abs(c[0, 2] - math.hypot(2.5, 1.5)) < 1e-12


def test_area_and_circularity_filters():
    p = D.hawkeye_params(720)
    assert len(D.blob_detect(square(30, 30, 5, 5, 3), p)) == 0     # area 4 < 7
    assert len(D.blob_detect(square(30, 30, 5, 5, 4), p)) == 1     # area 9
    big = square(60, 60, 5, 5, 33)                                  # area 1024
    assert len(D.blob_detect(big, p)) == 0
    line = np.zeros((20, 80), np.uint8)
    line[8:10, 5:70] = 255              # area 64, perimeter 130: ratio 0.048
    assert len(D.blob_detect(line, p)) == 0
    plo = D.hawkeye_params(360)
    assert len(D.blob_detect(square(30, 30, 5, 5, 2), plo)) == 1   # area 1


def test_hole_contour_rejected_by_colour():
    p = D.hawkeye_params(720)
    img = square(40, 40, 5, 5, 12)
    img[6:10, 6:10] = 0
    # outer: area 121, centroid (10.5, 10.5) -> cvRound (10, 10), a 255 pixel.
    # hole polygon: centroid (7.5, 7.5) -> (8, 8), a 0 pixel -> dropped
    kp = D.blob_detect(img, p)
    assert len(kp) == 1 and kp[0, 0] == np.float32(10.5)
    # a centred hole swallows the outer blob's own centroid: nothing survives
    img = square(40, 40, 5, 5, 12)
    img[8:14, 8:14] = 0
    assert len(D.blob_detect(img, p)) == 0


def test_close_blobs_merge_into_one_group():
    p = D.hawkeye_params(720)
    img = square(40, 40, 10, 10, 6) | square(40, 40, 10, 18, 6)
    c = D.find_blobs(img, p)
    assert c[:, 0].tolist() == [20.5, 12.5]         # reverse discovery order
    kp = D.blob_detect(img, p)
    assert len(kp) == 1
    want = (17 * 20.5 + 16 * 12.5) / 33
    assert abs(kp[0, 0] - want) < 1e-5 and kp[0, 1] == np.float32(12.5)
    far = square(40, 60, 10, 10, 6) | square(40, 60, 10, 30, 6)
    kp = D.blob_detect(far, p)
    assert kp[:, 0].tolist() == [32.5, 12.5]


# ---------------------------------------------------------------- MOG

def test_mog_first_frames_by_hand():
    m = D.Mog(4, 2)
    f = np.full((2, 4, 3), 100, np.uint8)
    assert not m.apply(f).any()                     # frame 1: all background
    mod = m.model()
    assert mod[0, 0, 1] == 1.0                      # weight 0.05 / 0.05
    assert mod[0, 0, 2:5].tolist() == [100, 100, 100]
    assert mod[0, 0, 5:8].tolist() == [900, 900, 900]
    sk0 = np.float32(np.float32(0.05) / (15. * 2 * math.sqrt(3.)))
    assert mod[0, 0, 0] == sk0 * np.float32(1.0 / np.float32(0.05))

    f2 = f.copy()
    f2[1, 2] = (250, 10, 10)
    mask = m.apply(f2)
    assert mask[1, 2] == 255 and mask.sum() == 255
    mod = m.model()
    px = mod[1 * 4 + 2]
    wsum = np.float32(1.0) + np.float32(0.0) + np.float32(0.05)
    scale = np.float32(1.0) / wsum
    assert px[0, 1] == np.float32(1.0) * scale
    assert px[1, 1] == np.float32(0.05) * scale
    assert px[1, 2:5].tolist() == [250, 10, 10]
    # a matched pixel: alpha = 1/2, diff = 0 -> var decays towards minVar
    q = mod[0]
    assert q[0, 5] == max(np.float32(900) + np.float32(0.5) *
                          (np.float32(0) - np.float32(900)), np.float32(225))
    assert D.lib().hkref_mog_alpha(1) == 1.0
    assert D.lib().hkref_mog_alpha(150) == np.float32(1. / 150)
    assert D.lib().hkref_mog_alpha(5000) == np.float32(1. / 200)


def test_mog_learns_a_static_scene_and_flags_a_moving_disc():
    rng = np.random.RandomState(3)
    h, w = 48, 64
    base = rng.randint(60, 180, size=(h, w, 3))
    m = D.Mog(w, h)
    yy, xx = np.mgrid[0:h, 0:w]
    for t in range(40):
        f = np.clip(base + rng.normal(0, 2, size=base.shape), 0,
                    255).astype(np.uint8)
        cx = 8 + t
        disc = (xx - cx)**2 + (yy - 24)**2 <= 16
        f[disc] = 255
        mask = m.apply(f)
        if t >= 2:
            # leading edge of the disc is new every frame -> foreground
            assert mask[24, cx + 4] == 255
            assert mask[5, 5] == 0
    assert set(np.unique(mask)) <= {0, 255}


# ---------------------------------------------------------------- real cv2

def _cv2_with_mog():
    cv2 = pytest.importorskip('cv2')
    if not hasattr(cv2, 'bgsegm'):
        pytest.skip('cv2 has no bgsegm (opencv_contrib) module')
    return cv2


def test_against_real_opencv_if_present():
    cv2 = _cv2_with_mog()
    print('cv2 version', cv2.__version__, '(reference pins 3.4.4)')
    rng = np.random.RandomState(9)
    h, w = 96, 128
    base = rng.randint(40, 200, size=(h, w, 3))
    ours, theirs = D.Mog(w, h), cv2.bgsegm.createBackgroundSubtractorMOG()
    sp = cv2.SimpleBlobDetector_Params()
    sp.filterByColor, sp.blobColor = True, 255
    sp.filterByCircularity, sp.minCircularity, sp.maxCircularity = \
        True, 0.3, 1.1
    sp.filterByInertia = sp.filterByConvexity = False
    sp.filterByArea, sp.minArea, sp.maxArea = True, 1.0, 150.0
    det = cv2.SimpleBlobDetector_create(sp)
    p = D.hawkeye_params(h)
    yy, xx = np.mgrid[0:h, 0:w]
    for t in range(60):
        f = np.clip(base + rng.normal(0, 3, size=base.shape), 0,
                    255).astype(np.uint8)
        f[(xx - 10 - 1.5 * t)**2 + (yy - 40)**2 <= 20] = 250
        a, b = ours.apply(f), theirs.apply(f)
        assert (a == b).mean() >= 0.999
        kp = det.detect(b)
        mine = D.blob_detect(b, p)
        assert len(kp) == len(mine)
        for k, q in zip(kp, mine):
            assert (np.float32(k.pt[0]), np.float32(k.pt[1]),
                    np.float32(k.size)) == tuple(q)


def test_detect_clip_threaded_equals_single_thread():
    rng = np.random.RandomState(3)
    T, H, W = 40, 61, 80
    base = rng.randint(60, 120, size=(H, W, 3))
    frames = np.empty((T, H, W, 3), np.uint8)
    for t in range(T):
        f = base + rng.randint(-6, 7, size=(H, W, 3))
        cx, cy = 10 + t, 20 + (t * 7) % 30
        f[max(0, cy - 4):cy + 4, max(0, cx - 4):cx + 4] = 250
        frames[t] = np.clip(f, 0, 255)
    bp = D.hawkeye_params(480)
    one = D.detect_clip(frames, bp, threads=1)
    many = D.detect_clip(frames, bp, threads=5, block=16)
    assert len(one) == len(many) == T
    assert sum(len(k) for k in one) > 0
    for a, b in zip(one, many):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
