"""Scratch probe: where the host time of the detection loop goes."""
import sys, time
import torch
sys.path.insert(0, '.')
from hawkeye_amd import synth
from hawkeye_amd.detect import EventDetector

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec, device=0).render(0, spec['frames'])
det = EventDetector(spec['width'], spec['height'], chunk=128, device=0)
for rep in range(3):
    det.mog.reset()
    torch.cuda.synchronize()
    t_sub = t_col = 0.0
    t0 = time.perf_counter()
    col_orig = det._collect
    def timed_collect(*p):
        global t_col
        a = time.perf_counter(); r = col_orig(*p); t_col += time.perf_counter() - a
        return r
    det._collect = timed_collect
    n = 0
    per_chunk = []
    for s in range(0, frames.shape[0], 128):
        a = time.perf_counter()
        out = det.submit(frames[s:s + 128])
        t_sub += time.perf_counter() - a
        per_chunk.append(time.perf_counter() - a)
        n += len(out)
    n += len(det.flush())
    det._collect = col_orig
    total = time.perf_counter() - t0
    per = sorted(per_chunk)
    print('per-chunk submit ms:', ' '.join('%.2f' % (v * 1e3) for v in per_chunk))
    print('total %.1f ms, submit (incl. collect) %.1f ms, collect %.1f ms, frames %d'
          % (total * 1e3, t_sub * 1e3, t_col * 1e3, n))
# GPU-only: all kernels without collecting in between
torch.cuda.synchronize()
masks = torch.empty((128, spec['height'], spec['width']), dtype=torch.uint8, device='cuda')
det.mog.reset()
det.mog.apply_batch(frames[256:384], out=masks)
for rep in range(3):
    det.blob[0].enqueue(masks)
    torch.cuda.synchronize()
    a = time.perf_counter()
    r = det.blob[0].finish()
    b = time.perf_counter()
    print('finish of a completed chunk: %.2f ms host, %d keypoints' % ((b - a) * 1e3, sum(len(x) for x in r)))
    det.blob[0].enqueue(masks)
    torch.cuda.synchronize()
    a = time.perf_counter()
    r = det.blob[0].finish_centers()
    b = time.perf_counter()
    print('finish_centers: %.2f ms host, %d centres' % ((b - a) * 1e3, sum(len(x) for x in r)))
