"""Step 3 of the benchmark clip's events: host threads vs the opt-in device
seeds (HK_STEP3_DEVICE), wall time of a single build."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
os.environ['HK_STEP3_STATS'] = '1'
import numpy as np                                      # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd.arcs import Clip                       # noqa: E402

d = np.load(os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), '_events_C.npz'))
F = int(d['frame'].max()) + 1
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nat.check(nat.lib.hk_set_host_threads(threads))
for rep in range(4):
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'],
                stream=0)
    t0 = time.perf_counter()
    clip.build_initial_arcs()
    print('threads %d device %s: step3 %.1f ms, arcs %d' % (
        threads, os.environ.get('HK_STEP3_DEVICE', '0'),
        (time.perf_counter() - t0) * 1e3, len(clip.arcs()[0])), flush=True)
    clip.close()
