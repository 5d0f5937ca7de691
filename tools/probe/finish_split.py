import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from hawkeye_amd import synth, _native as nat
from hawkeye_amd.batch import ClipScanner
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, 1024)
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
det = sc.detector
sc.detect(frames)
b0 = det.blob[0]
tc = []; tt = []
orig = nat.lib.hk_blob_finish
for s in range(256, 256 + 6 * 128, 128):
    det.mog.apply_batch_tiles(frames[s:s + 128], det._tiles[0])
    b0.enqueue_tiles(128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = nat.C.c_int()
    st = orig(b0._h, b0._counts.ctypes.data, b0._kp.ctypes.data, len(b0._kp), nat.C.byref(total))
    t1 = time.perf_counter()
    out = b0._split(b0._kp, total.value)
    t2 = time.perf_counter()
    tc.append((t1 - t0) * 1e6); tt.append((t2 - t1) * 1e6)
print('C call us:', ' '.join('%.0f' % t for t in tc)); print('split us:', ' '.join('%.0f' % t for t in tt))
