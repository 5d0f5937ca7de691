#!/usr/bin/env python3
"""What slows the detection thread inside a batch: the clip's detection timed
while N other threads (a) run step 3 of the same events over and over (CPU
only), (b) run steps 3-4 (CPU + small GPU kernels), (c) spin in C without the
interpreter lock (pure CPU pressure).  Not part of the product."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np                                      # noqa: E402
import torch                                            # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
events = sc.detect(frames)
nat.check(nat.lib.hk_set_host_threads(1))
stop = threading.Event()


def worker(kind):
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    while not stop.is_set():
        if kind == 'step3':
            res, clip = sc.build_arcs(events, em_stream=stream)
            clip.close()
        elif kind == 'step34':
            sc.refine_arcs(*sc.build_arcs(events, em_stream=stream))
        else:
            a = np.random.rand(400, 400)
            for _ in range(20):
                a = a @ a * 1e-3            # BLAS: no interpreter lock


def detect_ms(reps=12):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sc.detect(frames)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print('alone: %.2f ms per clip' % detect_ms())
for kind in ('step3', 'step34', 'blas'):
    for n in (8, 15):
        stop.clear()
        th = [threading.Thread(target=worker, args=(kind,), daemon=True)
              for _ in range(n)]
        for t in th:
            t.start()
        time.sleep(0.3)
        ms = detect_ms()
        stop.set()
        for t in th:
            t.join()
        print('%d threads of %s beside it: %.2f ms per clip' % (n, kind, ms))
