import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hawkeye_amd import synth
from hawkeye_amd.batch import ClipScanner
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
torch.cuda.synchronize()
for chunk in (120, 150, 180, 225, 300, 360):
    sc = ClipScanner(spec['width'], spec['height'], spec['fps'], chunk=chunk)
    for _ in range(3): sc.detect(frames)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sc.detect(frames); ts.append(time.perf_counter() - t0)
    print(chunk, 'lone detect ms: min %.2f median %.2f' % (min(ts) * 1e3, sorted(ts)[4] * 1e3))
    sc.close()
