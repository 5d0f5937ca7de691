"""Scratch probe: the blob chain on an all-background chunk (pure dispatch
cost of its kernels) against a chunk of the benchmark clip."""
import sys, time
import torch
sys.path.insert(0, '.')
from hawkeye_amd import synth
from hawkeye_amd.detect import EventDetector

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec, device=0).render(0, 640)
det = EventDetector(spec['width'], spec['height'], chunk=128, device=0)
for s in range(0, 512, 128):                     # model settled
    det.submit(frames[s:s + 128])
det.flush()
b = det.blob[0]
for name, prep in (('clip chunk', lambda: det.mog.apply_batch_tiles(frames[512:640], det._tiles[0])),
                   ('empty chunk (no foreground)', lambda: (det.mog.reset(), det.mog.apply_batch_tiles(frames[0:1].expand(128, -1, -1, -1).contiguous(), det._tiles[0])))):
    prep()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        b.enqueue_tiles(128)
        e1.record()
        kp = b.finish()
        ts.append(e0.elapsed_time(e1))
    print(name, 'blob chain %.0f us (min of 5), keypoints %d' % (min(ts) * 1e3, sum(len(k) for k in kp)))
