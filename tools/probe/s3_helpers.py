"""One build of the benchmark clip's step 3 on one thread, helped by N Python threads that call
hk_step3_help the way ClipScanner.scan_many's idle builders do: wall time of the build against
the library's own N+1 worker threads (what does helping cost beyond the threads themselves?).
    python3 tools/probe/s3_helpers.py"""
import os
import sys
import threading
import time

os.environ['HK_STEP3_STATS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np                                      # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd.arcs import Clip                       # noqa: E402

d = np.load(os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), '_events_C.npz'))
F = int(d['frame'].max()) + 1
SEEDS = int(os.environ.get('PROBE_HELP_SEEDS', '512'))

for helpers in (0, 1, 3, 7, 15):
    for rep in range(3):
        nat.check(nat.lib.hk_set_host_threads(1))
        clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
        done = threading.Event()
        calls = [0] * helpers

        def help_loop(k):
            while not done.is_set():
                if nat.lib.hk_step3_help(SEEDS):
                    calls[k] += 1
                else:
                    time.sleep(0.0002)

        ts = [threading.Thread(target=help_loop, args=(k,)) for k in range(helpers)]
        for t in ts:
            t.start()
        t0 = time.perf_counter()
        clip.build_initial_arcs()
        dt = (time.perf_counter() - t0) * 1e3
        done.set()
        for t in ts:
            t.join()
        print('%2d helpers: build %.1f ms (%d help calls)' % (helpers, dt, sum(calls)), flush=True)
        clip.close()
    nat.check(nat.lib.hk_set_host_threads(helpers + 1))
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
    t0 = time.perf_counter()
    clip.build_initial_arcs()
    print('%2d own threads: build %.1f ms' % (helpers + 1, (time.perf_counter() - t0) * 1e3), flush=True)
    clip.close()
