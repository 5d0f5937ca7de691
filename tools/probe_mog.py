"""Scratch timing probe for the MOG kernels (not part of the product)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd.detect import BackgroundSubtractorMOG

h, w = (1080, 1920) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
T = 64
g = torch.Generator(device='cuda').manual_seed(1)
base = torch.randint(40, 200, (h, w, 3), generator=g, device='cuda').float()
frames = torch.empty((T, h, w, 3), dtype=torch.uint8, device='cuda')
for i in range(T):
    f = base + 3.0 * torch.randn((h, w, 3), generator=g, device='cuda')
    f[200 + 5 * i:230 + 5 * i, 300 + 7 * i:330 + 7 * i] = 250
    frames[i] = f.clamp(0, 255).to(torch.uint8)
mog = BackgroundSubtractorMOG(w, h)
out = torch.empty((T, h, w), dtype=torch.uint8, device='cuda')
for name, fn in (('batch64', lambda: mog.apply_batch(frames, out=out)),
                 ('single x64', lambda: [mog.apply(frames[i], out=out[i]) for i in range(T)])):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fps = T / (ms * 1e-3)
    print(f'{name}: {ms:.3f} ms per {T} frames = {fps:.0f} fps; algorithmic 324 B/px -> {324*h*w*fps/1e12:.2f} TB/s')
