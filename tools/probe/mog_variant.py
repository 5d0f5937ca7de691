"""Scratch probe: time 128-frame MOG launches on clip-C frames (steady
state) for the HK_MOG_VARIANT in the environment.  argv[1] = 'static' uses a
static noisy scene (every wave on the fast path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hawkeye_amd import synth
from hawkeye_amd.detect import BackgroundSubtractorMOG
spec = synth.CLIPS['C_1080p60_9ball_noisy']
h, w = spec['height'], spec['width']
r = synth.ClipRenderer(spec)
frames = r.render(0, 512)
if len(sys.argv) > 1 and sys.argv[1] == 'static':
    g = torch.Generator(device='cuda').manual_seed(1)
    base = frames[0].float()
    for i in range(512):
        frames[i] = (base + 4.0 * torch.randn(base.shape, generator=g, device='cuda')).clamp(0, 255).to(torch.uint8)
mog = BackgroundSubtractorMOG(w, h)
out = torch.empty((128, h, w), dtype=torch.uint8, device='cuda')
ms = []
for i in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); mog.apply_batch(frames[128 * i:128 * i + 128], out=out); e1.record()
    torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
print('variant', os.environ.get('HK_MOG_VARIANT', '0'), sys.argv[1:], ' '.join('%.3f' % m for m in ms), 'ms; checksum', int(out.sum().item()))
