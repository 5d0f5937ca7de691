#!/usr/bin/env python3
"""Does detecting two clips at once (two ClipScanners, own streams and models,
one thread each) get more frames through the GPU than one after the other?
MOG launches fill every wave slot, but each launch ends with a tail of slow
waves that a second stream could fill.  Not part of the product."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
REPS = int(os.environ.get('REPS', '12'))


def run(n_threads):
    scanners = [ClipScanner(spec['width'], spec['height'], spec['fps'])
                for _ in range(n_threads)]
    for sc in scanners:
        sc.detect(frames)                               # warm
    torch.cuda.synchronize()
    go = threading.Barrier(n_threads + 1)

    def work(sc):
        torch.cuda.set_device(0)
        go.wait()
        for _ in range(REPS):
            sc.detect(frames)

    th = [threading.Thread(target=work, args=(sc,)) for sc in scanners]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    clips = REPS * n_threads
    print('%d detector(s): %.2f ms per clip, %.0f frames/s' % (
        n_threads, dt / clips * 1e3, clips * spec['frames'] / dt), flush=True)
    for sc in scanners:
        sc.close()


for n in (1, 2, 3, 1, 2):
    run(n)
