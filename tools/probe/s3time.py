import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hawkeye_amd import _native as nat
from hawkeye_amd.arcs import Clip
d = np.load('' + os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/_events_C.npz')
F = int(d['frame'].max()) + 1
nat.check(nat.lib.hk_set_host_threads(int(os.environ.get('T', '1'))))
best = 1e9
for rep in range(5):
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
    t0 = time.perf_counter()
    clip.build_initial_arcs()
    best = min(best, time.perf_counter() - t0)
    n = len(clip.arcs()[0])
    clip.close()
print(os.environ.get('HK_NATIVE_LIB', 'new')[-14:], 'threads', os.environ.get('T', '1'), 'step3 best %.1f ms, arcs %d' % (best * 1e3, n))
