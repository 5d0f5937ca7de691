#!/usr/bin/env python3
"""The detection thread inside a pipelined batch: how long each of its calls
takes (MOG launch, blob enqueue, finish) and how long it spends between them
(interpreter: waiting for the lock shows up there), as distributions."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
det = sc.detector
for _ in sc.scan_many([frames] * 8):
    pass
log = []        # (name, t0, t1)


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        log.append((name, t0, time.perf_counter()))
        return r
    return wrapper


det.mog.apply_batch_tiles = timed('mog launch', det.mog.apply_batch_tiles)
for b in det.blob:
    b.enqueue_tiles = timed('blob enqueue', b.enqueue_tiles)
    b._finish_into = timed('finish', b._finish_into)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for _ in sc.scan_many([frames] * K):
    n += 1
total = time.perf_counter() - t0
print('batch of {}: {:.2f} ms per clip'.format(K, total / K * 1e3))
log.sort(key=lambda r: r[1])
# steady state: the middle half of the calls
lo, hi = len(log) // 4, 3 * len(log) // 4
span = log[hi][1] - log[lo][1]
cats = {}
for i in range(lo, hi):
    name, a, b = log[i]
    cats.setdefault(name, []).append(b - a)
    cats.setdefault('between calls', []).append(log[i + 1][1] - b)
chunks = len(cats['mog launch'])
print('steady state: {} chunks in {:.1f} ms = {:.1f} us per chunk'.format(
    chunks, span * 1e3, span / chunks * 1e6))
for name, v in cats.items():
    v = np.array(v) * 1e6
    print('  {:14s} n {:5d}  sum/chunk {:7.1f} us   pct 50/90/99 {:7.1f} {:7.1f} {:7.1f}  '
          'max {:8.1f}   in calls > 300 us: {:6.1f} us/chunk'.format(
              name, len(v), v.sum() / chunks, *np.percentile(v, [50, 90, 99]),
              v.max(), v[v > 300].sum() / chunks))

# cycles (MOG launch to MOG launch) that took much longer than a launch: where
launches = [i for i in range(lo, hi) if log[i][0] == 'mog launch']
late = []
for a, b in zip(launches[:-1], launches[1:]):
    cyc = log[b][1] - log[a][1]
    if cyc > 0.0009:
        parts = {}
        for i in range(a, b):
            parts[log[i][0]] = parts.get(log[i][0], 0) + log[i][2] - log[i][1]
            parts['between'] = parts.get('between', 0) + log[i + 1][1] - log[i][2]
        late.append((cyc, parts))
print('cycles > 900 us: {} of {}, {:.1f} us per chunk in excess of 500'.format(
    len(late), len(launches), sum(c - 0.0005 for c, _ in late) / len(launches) * 1e6))
for cyc, parts in sorted(late, key=lambda r: -r[0])[:12]:
    print('  {:7.0f} us: '.format(cyc * 1e6) + ', '.join(
        '{} {:.0f}'.format(k, v * 1e6) for k, v in parts.items()))
