#!/usr/bin/env python3
"""Throughput of consecutive pipelined batches in one fresh process (does the
box warm up, and how long does it take?).  Not part of the product."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

n = 20
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
start = time.perf_counter()
for _ in range(5):
    sc.scan(frames)
for rep in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    det = 0.0
    for r in sc.scan_many((frames for _ in range(n)), count=n):
        det += r.stage_seconds['detect']
    dt = time.perf_counter() - t0
    print('t={:5.1f}s batch {}: {:.0f} frames/s, detect {:.2f} ms per clip'
          .format(time.perf_counter() - start, rep, n * spec['frames'] / dt,
                  det / n * 1e3), flush=True)
