"""N independent one-thread builds of the benchmark clip's step 3 at the same time (own clip
each): does a build slow down because the other cores are busy, or only when the threads share
a clip (tools/probe/s3_threads.py)?
    python3 tools/probe/s3_independent.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np                                      # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd.arcs import Clip                       # noqa: E402

d = np.load(os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), '_events_C.npz'))
F = int(d['frame'].max()) + 1
nat.check(nat.lib.hk_set_host_threads(1))


def one(out, k, barrier):
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
    barrier.wait()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    clip.build_initial_arcs()
    out[k] = ((time.perf_counter() - t0) * 1e3, (time.thread_time() - c0) * 1e3)
    clip.close()


for n in (1, 2, 4, 8, 12, 16, 1):
    for rep in range(2):
        out = [None] * n
        barrier = threading.Barrier(n)
        ts = [threading.Thread(target=one, args=(out, k, barrier)) for k in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        print('%2d builds at once: wall per build %.1f ms (max %.1f), cpu %.1f ms' % (
            n, sum(o[0] for o in out) / n, max(o[0] for o in out),
            sum(o[1] for o in out) / n), flush=True)
