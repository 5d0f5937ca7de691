#!/usr/bin/env python3
"""Randomised parity run of the blob detector on the GPU: random binary masks
(thresholded smooth noise of random scale, pure noise of random density,
discs, rings and specks; random sizes, odd ones included) through BOTH
hand-overs -- byte masks (hk_blob_enqueue) and, where frames are made of
32 x 4 tiles, the MOG kernel's 1-bit tile format (hk_blob_enqueue_tiles, tiles
packed here with numpy) -- per-contour centres with and without the colour
filter (without it every traced hole shows) compared bit for bit with the C
oracle.  Stops at the first mismatch.  Not collected by pytest.

    python tests/fuzz_blob.py [--seconds 120] [--seed 1]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch
from scipy import ndimage

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd.detect import SimpleBlobDetector, hawkeye_blob_params  # noqa: E402
from oracle import detect_ref as D                          # noqa: E402


def random_mask(rng, h, w):
    kind = rng.randint(0, 4)
    if kind == 0:
        size = rng.uniform(0.6, 6.0)
        f = ndimage.gaussian_filter(rng.normal(size=(h, w)), size)
        m = f > np.quantile(f, rng.uniform(0.2, 0.9))
    elif kind == 1:
        m = rng.uniform(size=(h, w)) < rng.choice([0.02, 0.1, 0.3, 0.5, 0.7,
                                                   0.9])
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        m = np.zeros((h, w), bool)
        for _ in range(rng.randint(1, 25)):
            cx, cy = rng.uniform(-5, w + 5), rng.uniform(-5, h + 5)
            r = rng.uniform(1, 16)
            d2 = (xx - cx) ** 2 + (yy - cy) ** 2
            m |= d2 <= r * r
            if rng.uniform() < 0.4:         # a ring: a hole inside
                ri = r * rng.uniform(0.2, 0.8)
                m &= ~(d2 < ri * ri)
        if kind == 3:
            m ^= rng.uniform(size=(h, w)) < 0.01
    return (m * 255).astype(np.uint8)


def pack_tiles(masks):
    """uint8 [T][H][W] -> uint32 [T][H/4][W/32][4]: per 32x4 tile the even
    pixels of rows 0-1 and rows 2-3, then the odd pixels (16 bits per row)."""
    t, h, w = masks.shape
    b = (masks > 0).reshape(t, h // 4, 4, w // 32, 16, 2).astype(np.uint32)
    weights = (1 << np.arange(16, dtype=np.uint32))
    rows = (b * weights[None, None, None, None, :, None]).sum(axis=4)
    # rows[t][ty][r][tx][parity]
    out = np.empty((t, h // 4, w // 32, 4), np.uint32)
    out[..., 0] = rows[:, :, 0, :, 0] | (rows[:, :, 1, :, 0] << 16)
    out[..., 1] = rows[:, :, 2, :, 0] | (rows[:, :, 3, :, 0] << 16)
    out[..., 2] = rows[:, :, 0, :, 1] | (rows[:, :, 1, :, 1] << 16)
    out[..., 3] = rows[:, :, 2, :, 1] | (rows[:, :, 3, :, 1] << 16)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=120)
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0 = said = time.time()
    cases = contours = tiled = 0
    while time.time() - t0 < args.seconds:
        w = int(rng.choice([rng.randint(33, 400), 64, 96, 320, 640, 641]))
        h = int(rng.choice([rng.randint(8, 300), 64, 128, 132, 200]))
        tiles_ok = rng.uniform() < 0.5
        if tiles_ok:
            w, h = max(32, w // 32 * 32), max(4, h // 4 * 4)
        nf = int(rng.randint(1, 6))
        masks = np.stack([random_mask(rng, h, w) for _ in range(nf)])
        ph = int(rng.choice([h, 480, 240]))
        by_color = bool(rng.randint(0, 2))
        p = D.hawkeye_params(ph)
        p.filterByColor = 1 if by_color else 0
        want = [D.find_blobs(m, p) for m in masks]
        params = hawkeye_blob_params(ph)
        params.filterByColor = 1 if by_color else 0
        det = SimpleBlobDetector(w, h, params=params, max_frames=nf)
        dev = torch.from_numpy(masks).cuda()
        det.enqueue(dev)
        got = [det.finish_centers()]
        if tiles_ok:
            packed = torch.from_numpy(pack_tiles(masks).view(np.int32)).cuda()
            torch.cuda.synchronize()
            _copy(det.tiles_buffer(), packed)
            det.enqueue_tiles(nf)
            got.append(det.finish_centers())
            tiled += 1
        for which, g in enumerate(got):
            for f in range(nf):
                if g[f].shape != want[f].shape or not np.array_equal(
                        g[f].view(np.uint64), want[f].view(np.uint64)):
                    np.save('/tmp/fuzz_blob_fail.npy', masks)
                    print('MISMATCH case', cases, 'path',
                          'tiles' if which else 'bytes', 'frame', f,
                          'size', w, h, 'params_height', ph, 'by_color',
                          by_color, 'got', len(g[f]), 'want', len(want[f]))
                    sys.exit(1)
        contours += sum(len(x) for x in want)
        det.close()
        cases += 1
        if time.time() - said > 30:     # a silent GPU run is taken to be hung
            print('...', cases, 'cases', flush=True)
            said = time.time()
    print('fuzz_blob: {} cases ({} also through the tile hand-over), {} '
          'contours that passed the filters, no difference'.format(
              cases, tiled, contours))


def _copy(dst_ptr, src):
    hip = ctypes.CDLL('libamdhip64.so')
    rc = hip.hipMemcpy(ctypes.c_void_p(dst_ptr),
                       ctypes.c_void_p(src.data_ptr()),
                       ctypes.c_size_t(src.numel() * 4), 3)
    if rc != 0:
        raise RuntimeError('hipMemcpy failed: {}'.format(rc))
    return 0


if __name__ == '__main__':
    main()
