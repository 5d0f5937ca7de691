#!/usr/bin/env python3
"""Where the detection thread's time goes, call by call (single thread, the
benchmark clip resident in HBM): per chunk the host time inside the MOG launch
call, the blob enqueue, the C finish call and the Python split of its result.
Not part of the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
det = sc.detector
for _ in range(3):
    sc.detect(frames)

acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrapper


det.mog.apply_batch_tiles = timed('mog launch', det.mog.apply_batch_tiles)
for b in det.blob:
    b.enqueue_tiles = timed('blob enqueue', b.enqueue_tiles)
    b._split = timed('python split', b._split)
    b._finish_into = timed('finish (C call + split)', b._finish_into)
reps = 10
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ev = sc.detect(frames)
total = time.perf_counter() - t0
chunks = reps * ((spec['frames'] + 127) // 128)
print('detect: {:.2f} ms per clip, {:.1f} us per chunk'.format(
    total / reps * 1e3, total / chunks * 1e6))
for k, v in acc.items():
    print('  {:28s} {:8.1f} us per chunk'.format(k, v / chunks * 1e6))
