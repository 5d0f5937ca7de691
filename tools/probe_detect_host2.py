"""Scratch probe: per-call host times inside EventDetector.submit()."""
import sys, time
import torch
sys.path.insert(0, '.')
from hawkeye_amd import synth
from hawkeye_amd import detect as D

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec, device=0).render(0, spec['frames'])
det = D.EventDetector(spec['width'], spec['height'], chunk=128, device=0)
log = []


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        log.append((tag, (time.perf_counter() - t) * 1e3))
        return r
    setattr(obj, name, g)


wrap(det.mog, 'apply_batch_tiles', 'mog')
for b in det.blob:
    wrap(b, 'enqueue_tiles', 'enq')
    wrap(b, 'finish', 'fin')
for rep in range(6):
    det.mog.reset()
    torch.cuda.synchronize()
    del log[:]
    t0 = time.perf_counter()
    for s in range(0, frames.shape[0], 128):
        det.submit(frames[s:s + 128])
    det.flush()
    total = (time.perf_counter() - t0) * 1e3
    big = [(t, '%.2f' % v) for t, v in log if v > 1.5]
    tot = {}
    for t, v in log:
        tot[t] = tot.get(t, 0) + v
    print('total %.1f ms' % total, {k: round(v, 1) for k, v in tot.items()}, 'calls > 1.5 ms:', big)
