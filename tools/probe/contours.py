#!/usr/bin/env python3
"""What the border tracer is asked to do on the benchmark clip: per frame the
number of border-start candidates (the scan kernel's local rule), the
contours OpenCV's findContours returns (C oracle) and their lengths.
Not part of the product.   python3 tools/probe/contours.py [first] [count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.detect import BackgroundSubtractorMOG  # noqa: E402
from oracle import detect_ref as D                      # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 900
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, first + count)
mog = BackgroundSubtractorMOG(spec['width'], spec['height'])
masks = None
for s in range(0, first + count, 128):
    out = mog.apply_batch(frames[s:s + 128])
    if s + 128 > first:
        lo = max(first - s, 0)
        part = out[lo:].cpu().numpy()
        masks = part if masks is None else np.concatenate([masks, part])
masks = masks[:count]
lens_all, cands_all = [], []
for m in masks:
    fg = np.pad(m > 0, 1)
    c = fg[1:-1, 1:-1]
    W, NW, N, NE = fg[1:-1, :-2], fg[:-2, :-2], fg[:-2, 1:-1], fg[:-2, 2:]
    outer = c & ~W & ~NW & ~N & ~NE
    hole = ~c & W & N
    cands_all.append((int(outer.sum()), int(hole.sum())))
    cs = D.find_contours(m)
    lens_all.append([(h, len(p)) for h, p in cs])
print('frames', len(masks), 'foreground px per frame',
      int((masks > 0).sum() / len(masks)))
print('candidates per frame: outer %.1f hole %.1f' % tuple(
    np.mean(cands_all, axis=0)))
ln = np.array([l for f in lens_all for _, l in f])
hl = np.array([h for f in lens_all for h, _ in f])
print('contours per frame %.1f (holes %.1f)' % (len(ln) / len(masks),
                                                 hl.sum() / len(masks)))
for lo, hi in ((1, 1), (2, 4), (5, 16), (17, 64), (65, 207), (208, 10**9)):
    sel = (ln >= lo) & (ln <= hi)
    print('  length %4d..%-10d: %7.1f per frame, %8.1f points per frame' % (
        lo, hi, sel.sum() / len(masks), ln[sel].sum() / len(masks)))
