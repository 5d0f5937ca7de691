import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hawkeye_amd.arcs import Clip
d = np.load(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/events_C_1080p60_9ball_noisy.npz')
F = int(d['frame'].max()) + 1
for rep in range(4):
    clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
    t0 = time.perf_counter(); clip.build_initial_arcs(); t1 = time.perf_counter()
    print('step3 %.1f ms, arcs %d' % ((t1 - t0) * 1e3, len(clip.arcs()[0])))
