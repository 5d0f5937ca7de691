"""Scratch probe: render config C frames, run MOG batch + blob detect, print
timings and event statistics (not part of the product)."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import synth
from hawkeye_amd.detect import BackgroundSubtractorMOG, SimpleBlobDetector

name = sys.argv[1] if len(sys.argv) > 1 else 'C_1080p60_9ball_noisy'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spec = synth.CLIPS[name]
h, w = spec['height'], spec['width']
r = synth.ClipRenderer(spec)
t0 = time.time(); frames = r.render(0, T); torch.cuda.synchronize(); print('render', time.time() - t0)
mog = BackgroundSubtractorMOG(w, h)
det = SimpleBlobDetector(w, h, max_frames=128)
masks = torch.empty((T, h, w), dtype=torch.uint8, device='cuda')
def ev(): 
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(2):
    mog.reset()
    a = ev(); mog.apply_batch(frames, out=masks); b = ev(); torch.cuda.synchronize()
    t0 = time.time(); kps = det.detect(masks); t1 = time.time()
    print(f'rep {rep}: mog {a.elapsed_time(b):.3f} ms / {T} frames; blob (incl host) {(t1-t0)*1e3:.3f} ms')
cnt = [len(k) for k in kps]
print('fg fraction per frame', [float((masks[i] == 255).float().mean()) for i in (1, 2, 10, 100, T - 1)])
print('keypoints/frame: min', min(cnt), 'max', max(cnt), 'mean', sum(cnt) / len(cnt))
print('frame 100 kps', kps[100][:12])
print('discs frame 100', r.discs[100][:12].cpu().numpy())
