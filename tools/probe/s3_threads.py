"""Step 3 of the benchmark clip's events on 1, 2, 4, 8, 16 threads: wall time, seed time summed
over the threads, seeds run and growth steps (what does running ahead of the committer cost?).  PROBE_FRAMES=n: the first n frames only.
    python3 tools/probe/s3_threads.py"""
import os
import sys
import time

os.environ['HK_STEP3_STATS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np                                      # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd.arcs import Clip                       # noqa: E402

d = np.load(os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), '_events_C.npz'))
F = int(os.environ.get('PROBE_FRAMES', int(d['frame'].max()) + 1))
keep = d['frame'] < F
TILE = int(os.environ.get('PROBE_TILE', '1'))   # the clip repeated in time
fr = np.concatenate([d['frame'][keep] + k * F for k in range(TILE)])
xx, yy, rr = (np.tile(d[k][keep], TILE) for k in 'xyr')
for threads in (1, 2, 4, 8, 16):
    nat.check(nat.lib.hk_set_host_threads(threads))
    for rep in range(3):
        clip = Clip(1920, 1080, 60.0, F * TILE, fr, xx, yy, rr, stream=0)
        t0 = time.perf_counter()
        clip.build_initial_arcs()
        print('threads %d: step3 %.1f ms' % (threads, (time.perf_counter() - t0) * 1e3),
              flush=True)
        clip.close()
