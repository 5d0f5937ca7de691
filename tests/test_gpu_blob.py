"""
GPU parity: the HIP blob detector (through the C ABI) against the C oracle's
restatement of cv2.SimpleBlobDetector, bit for bit, on the same masks:
per-contour centres (float64 x, y, radius) and final keypoints (float32).
"""
import numpy as np
import pytest
from scipy import ndimage

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def mask_shapes(rng, h, w, n_discs=14, specks=30, big=True):
    """Discs/ellipses of many sizes, rings, touching pairs, edge blobs, a
    large 'juggler' region and single-pixel specks."""
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    for _ in range(n_discs):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        a, b = rng.uniform(1.0, 14.0), rng.uniform(1.0, 14.0)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        e = (u / a)**2 + (v / b)**2 <= 1
        if rng.uniform() < 0.3:
            e &= ~((u / (a * 0.5))**2 + (v / (b * 0.5))**2 <= 1)    # ring
        m |= e
    if big:
        m[h // 3:h // 3 + h // 2, w // 2:w // 2 + w // 5] = True
        m[h // 3 + 5:h // 3 + 9, w // 2 + 3:w // 2 + 9] = False     # hole
    for _ in range(specks):
        m[rng.randint(0, h), rng.randint(0, w)] ^= True
    return m.astype(np.uint8) * 255


def mask_smooth_noise(rng, h, w, size, thr):
    f = ndimage.uniform_filter(rng.uniform(size=(h, w)), size)
    return (f > thr).astype(np.uint8) * 255


def oracle_results(masks, params_height, by_color=True):
    from oracle import detect_ref as D
    p = D.hawkeye_params(params_height)
    p.filterByColor = 1 if by_color else 0
    return ([D.find_blobs(m, p) for m in masks],
            [D.blob_detect(m, p) for m in masks])


def check_against_oracle(masks, params_height=None, by_color=True):
    """by_color=False: without the colour test hole contours are not dropped
    (their centroid lies on background), so every traced hole shows."""
    from hawkeye_amd.detect import SimpleBlobDetector, hawkeye_blob_params
    t, h, w = masks.shape
    ph = h if params_height is None else params_height
    want_c, want_k = oracle_results(masks, ph, by_color)
    params = hawkeye_blob_params(ph)
    params.filterByColor = 1 if by_color else 0
    det = SimpleBlobDetector(w, h, params=params,
                             max_frames=max(1, min(t, 16)))
    dev = torch.from_numpy(masks).cuda()
    got_c = []
    for s in range(0, t, det.max_frames):
        det.enqueue(dev[s:s + det.max_frames])
        got_c.extend(det.finish_centers())
    got_k = det.detect(dev)
    n_c = n_k = 0
    for f in range(t):
        assert got_c[f].shape == want_c[f].shape, \
            'frame {}: {} centres, oracle {}'.format(f, len(got_c[f]),
                                                     len(want_c[f]))
        assert np.array_equal(got_c[f].view(np.uint64),
                              want_c[f].view(np.uint64)), 'frame {}'.format(f)
        assert got_k[f].shape == want_k[f].shape
        assert np.array_equal(got_k[f].view(np.uint32),
                              want_k[f].view(np.uint32)), 'frame {}'.format(f)
        n_c += len(want_c[f])
        n_k += len(want_k[f])
    return n_c, n_k


@pytest.mark.parametrize('h,w', [(96, 128), (61, 83), (480, 640), (70, 8200)])
def test_shapes_match_oracle(h, w):
    rng = np.random.RandomState(h + w)
    masks = np.stack([mask_shapes(rng, h, w) for _ in range(12)])
    n_c, n_k = check_against_oracle(masks)
    assert n_c > 10 and n_k > 5


def test_low_res_parameter_block():
    rng = np.random.RandomState(4)
    masks = np.stack([mask_shapes(rng, 120, 160, n_discs=10) for _ in
                      range(6)])
    n_c, n_k = check_against_oracle(masks, params_height=360)
    assert n_c > 10


@pytest.mark.parametrize('size,thr', [(3, 0.5), (5, 0.5), (7, 0.52),
                                      (3, 0.62), (9, 0.5), (4, 0.4)])
def test_messy_topology_matches_oracle(size, thr):
    """Thresholded smoothed noise: nested holes, islands in holes, diagonal
    links, one-pixel bridges -- everything border following can meet."""
    rng = np.random.RandomState(size * 100 + int(thr * 100))
    masks = np.stack([mask_smooth_noise(rng, 90, 117, size, thr)
                      for _ in range(8)])
    check_against_oracle(masks)
    check_against_oracle(masks, params_height=300)


def test_pure_noise_and_extremes():
    rng = np.random.RandomState(17)
    masks = np.stack(
        [(rng.uniform(size=(64, 80)) < d).astype(np.uint8) * 255
         for d in (0.02, 0.1, 0.3, 0.5, 0.7, 0.9)] +
        [np.zeros((64, 80), np.uint8), np.full((64, 80), 255, np.uint8)])
    check_against_oracle(masks)
    check_against_oracle(masks, params_height=300)


def test_dense_noise_grows_the_work_lists():
    """More border candidates in a band of rows, and more surviving contours,
    than the work lists start with (2048 per band, 128 per frame): the
    detector grows them and runs the batch again; a sparse batch afterwards
    still works."""
    rng = np.random.RandomState(31)
    h, w = 160, 1920
    dense = np.stack(
        [(rng.uniform(size=(h, w)) < d).astype(np.uint8) * 255
         for d in (0.05, 0.2)] +
        [mask_smooth_noise(rng, h, w, 5, 0.52)])
    check_against_oracle(dense)
    from hawkeye_amd.detect import SimpleBlobDetector, hawkeye_blob_params
    from oracle import detect_ref as D
    det = SimpleBlobDetector(w, h, params=hawkeye_blob_params(h), max_frames=2)
    sparse = mask_shapes(rng, h, w, n_discs=10, specks=5, big=False)
    for m in (dense[1], sparse, dense[2], sparse):
        got = det.detect(torch.from_numpy(m[None]).cuda())[0]
        want = D.blob_detect(m, D.hawkeye_params(h))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_holes_that_start_on_the_first_row_of_a_band():
    """The border tracer works on bands of 64 rows out of an LDS window.  A
    hole border steps up to row y0 - 1 on its first move and reads row y0 - 2
    there: holes whose first pixel lies on rows 64 / 128 (first row of a
    band), plain and as the lower part of a hole that starts further up (the
    trace then climbs out of the band and must retire)."""
    h, w = 200, 192
    masks = np.zeros((4, h, w), np.uint8)
    for f, y0 in enumerate((64, 128, 64, 128)):
        m = masks[f]
        for i, x0 in enumerate((10, 50, 90, 130)):
            m[y0 - 4 - i:y0 + 14, x0:x0 + 24] = 255     # block
            m[y0:y0 + 5 + i, x0 + 4:x0 + 10 + i] = 0     # hole, top row y0
        if f >= 2:
            # second arm of the hole of block 1, reaching 3 rows higher:
            # the candidate at (x0 + 4, y0) is no longer the hole's first pixel
            x0 = 50
            m[y0 + 3, x0 + 4:x0 + 18] = 0
            m[y0 - 3:y0 + 4, x0 + 17] = 0
            # diagonal chain of holes across the band border (8-connected
            # foreground between them)
            x0 = 130
            for k in range(5):
                m[y0 - 3 + k, x0 + 14 + k] = 0
        # a ring whose hole starts exactly one row above the band
        yy, xx = np.mgrid[0:h, 0:w]
        r2 = (xx - 170)**2 + (yy - (y0 + 5))**2
        m[(r2 <= 81) & (r2 >= 30)] = 255
    # (the parameter block of scan heights >= 480: areas up to 1000)
    n_c, n_k = check_against_oracle(masks, params_height=480)
    assert n_c >= 8
    n_all, _ = check_against_oracle(masks, params_height=480, by_color=False)
    assert n_all >= n_c + 12            # the holes are contours of their own
    # the same picture shifted by one and two rows: starts on the last rows of
    # the band above
    for shift in (1, 2, 63):
        check_against_oracle(np.roll(masks, -shift, axis=1),
                             params_height=480, by_color=False)


def test_full_hd_sparse_blobs():
    rng = np.random.RandomState(23)
    h, w = 1080, 1920
    masks = np.stack([mask_shapes(rng, h, w, n_discs=40, specks=200)
                      for _ in range(3)])
    n_c, n_k = check_against_oracle(masks)
    assert n_k >= 20


def test_rejects_non_binary_mask_and_bad_params():
    from hawkeye_amd import _native as nat
    from hawkeye_amd.detect import SimpleBlobDetector, hawkeye_blob_params
    det = SimpleBlobDetector(64, 48)
    m = torch.zeros((1, 48, 64), dtype=torch.uint8, device='cuda')
    m[0, 10:20, 10:20] = 255
    assert len(det.detect(m)[0]) == 1
    m[0, 30, 30] = 128
    with pytest.raises(nat.NativeError, match='other than 0 and 255'):
        det.detect(m)
    p = hawkeye_blob_params(720)
    p.filterByInertia = 1
    with pytest.raises(nat.NativeError, match='inertia'):
        SimpleBlobDetector(64, 48, params=p)
    p = hawkeye_blob_params(720)
    p.maxArea = 50000.0
    with pytest.raises(nat.NativeError, match='points'):
        SimpleBlobDetector(64, 48, params=p)


def test_shapes_that_stress_the_candidate_pruning():
    """The scan kernels drop border-start candidates that a look at the row
    above shows not to be starts (prune_candidates).  Shapes made for that
    rule: combs with teeth up (many outer candidates in one component) and
    down (many hole candidates in the outer background), U shapes, rings
    with a notch, nested rings, a ring whose hole starts under a run that is
    only partly covered, blobs cut by every image edge, a checkerboard, and
    one-pixel-wide diagonals.  With and without the colour filter, at a size
    that goes through the tile hand-over as well."""
    h, w = 128, 160
    m = np.zeros((6, h, w), np.uint8)
    # 0: combs
    m[0, 20:24, 10:70] = 255
    for x in range(10, 70, 4):
        m[0, 12:20, x:x + 2] = 255          # teeth up
    m[0, 60:64, 10:70] = 255
    for x in range(10, 70, 6):
        m[0, 64:72, x:x + 3] = 255          # teeth down
    m[0, 90:100, 90:94] = 255               # U
    m[0, 90:100, 100:104] = 255
    m[0, 98:102, 90:104] = 255
    # 1: rings, notched rings, nested rings
    yy, xx = np.mgrid[0:h, 0:w]
    for cx, cy, ro, ri in ((30, 30, 12, 6), (80, 30, 14, 9), (125, 35, 16, 4)):
        d2 = (xx - cx) ** 2 + (yy - cy) ** 2
        m[1][(d2 <= ro * ro) & (d2 >= ri * ri)] = 255
    m[1, 16:24, 84:87] = 0                  # notch through the top of a ring
    d2 = (xx - 60) ** 2 + (yy - 90) ** 2
    m[1][(d2 <= 30 * 30) & (d2 >= 24 * 24)] = 255
    m[1][(d2 <= 18 * 18) & (d2 >= 12 * 12)] = 255
    m[1][d2 <= 5 * 5] = 255
    # 2: holes whose first row lies under a partly covered run
    m[2, 20:40, 20:60] = 255
    m[2, 25:35, 30:50] = 0
    m[2, 24, 44:50] = 0                     # the hole's top is a step
    m[2, 60:80, 20:60] = 255
    m[2, 65:75, 30:50] = 0
    m[2, 60:65, 40] = 0                     # a slit from the hole to outside
    m[2, 90:110, 100:140] = 255
    m[2, 95:105, 110:130] = 0
    m[2, 100, 105:112] = 0                  # a side pocket, lower than the top
    # 3: blobs cut by the edges
    for cx, cy in ((0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1),
                   (w // 2, 0), (w - 1, h // 2), (0, h // 2), (w // 2, h - 1)):
        m[3][(xx - cx) ** 2 + (yy - cy) ** 2 <= 81] = 255
    m[3, 40:50, w - 6:w] = 255
    m[3, 43:47, w - 3:w] = 0                # a bay open to the right edge
    # 4: checkerboard and diagonals
    m[4][((xx // 3 + yy // 3) % 2 == 0) & (xx < 60) & (yy < 60)] = 255
    for k in range(40):
        m[4, 70 + k, 20 + k] = 255
        m[4, 70 + k, 100 - k] = 255
        m[4, 70 + k, 101 - k] = 255
    # 5: everything at once, shifted by 31/33 pixels (word boundaries)
    m[5] = np.roll(np.roll(m[0] | m[1], 31, axis=1), 3, axis=0) | \
        np.roll(m[2], 33, axis=1)
    for by_color in (True, False):
        n_c, _ = check_against_oracle(m, params_height=480, by_color=by_color)
        assert n_c > 10
    # the same through the 1-bit tile hand-over
    import ctypes
    from hawkeye_amd.detect import SimpleBlobDetector, hawkeye_blob_params
    from oracle import detect_ref as D
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'fuzz_blob', os.path.join(os.path.dirname(__file__), 'fuzz_blob.py'))
    fuzz_blob = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz_blob)
    pack_tiles = fuzz_blob.pack_tiles
    for by_color in (True, False):
        p = D.hawkeye_params(480)
        p.filterByColor = 1 if by_color else 0
        want = [D.find_blobs(f, p) for f in m]
        params = hawkeye_blob_params(480)
        params.filterByColor = 1 if by_color else 0
        det = SimpleBlobDetector(w, h, params=params, max_frames=len(m))
        packed = torch.from_numpy(pack_tiles(m).view(np.int32)).cuda()
        torch.cuda.synchronize()
        hip = ctypes.CDLL('libamdhip64.so')
        assert hip.hipMemcpy(ctypes.c_void_p(det.tiles_buffer()),
                             ctypes.c_void_p(packed.data_ptr()),
                             ctypes.c_size_t(packed.numel() * 4), 3) == 0
        det.enqueue_tiles(len(m))
        got = det.finish_centers()
        for f in range(len(m)):
            assert got[f].shape == want[f].shape, (by_color, f)
            assert np.array_equal(got[f].view(np.uint64),
                                  want[f].view(np.uint64)), (by_color, f)
        det.close()
