"""Step 3 with the HK_S3_PROF build (tools/probe/libhk_s3prof.so, tools/probe/build_probe_libs.sh
s3prof): where the cycles of a build go, section by section (HK_S3_PHASE=k times one section
alone), and how its trials end.  Events of the benchmark clip from tools/_events_C.npz
(tools/dump_events.py).
    python3 tools/probe/s3prof.py [threads]"""
import os
import sys
import time

os.environ['HK_NATIVE_LIB'] = os.path.join(os.path.dirname(
    os.path.abspath(__file__)), 'libhk_s3prof.so')
os.environ['HK_STEP3_STATS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import numpy as np                                      # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd.arcs import Clip                       # noqa: E402

d = np.load(os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), '_events_C.npz'))
F = int(d['frame'].max()) + 1
nat.check(nat.lib.hk_set_host_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 1))
for rep in range(3):
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'],
                stream=0)
    t0 = time.perf_counter()
    clip.build_initial_arcs()
    print('step3 %.1f ms, arcs %d' % ((time.perf_counter() - t0) * 1e3,
                                      len(clip.arcs()[0])), flush=True)
    clip.close()
