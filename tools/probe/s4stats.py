import os, sys, time
os.environ['HK_STEP4_STATS'] = '1'
sys.path.insert(0, '/root/repo')
import numpy as np
from hawkeye_amd.arcs import Clip
d = np.load('/root/repo/tools/_events_C.npz')
F = int(d['frame'].max()) + 1
for rep in range(3):
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
    clip.build_initial_arcs()
    t0 = time.perf_counter(); clip.em_optimize(); print('step4 %.2f ms' % ((time.perf_counter()-t0)*1e3), flush=True)
    clip.close()
