"""Scratch probe: where the detection thread's time goes inside scan_many."""
import sys, time
import torch
sys.path.insert(0, '.')
from hawkeye_amd import synth
from hawkeye_amd.batch import ClipScanner

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec, device=0).render(0, spec['frames'])
torch.cuda.synchronize()
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
det = sc.detector
tot = {}


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        tot[tag] = tot.get(tag, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)


wrap(det.mog, 'apply_batch_tiles', 'mog launch')
wrap(det.mog, 'reset', 'reset')
for b in det.blob:
    wrap(b, 'enqueue_tiles', 'blob enqueue')
    wrap(b, 'finish', 'blob finish (wait + grouping)')
    wrap(b, '_split', 'split (python)')
wrap(sc, 'detect', 'detect total')
for _ in range(3):
    sc.scan(frames)
for K in (1, 40):
    tot.clear()
    t0 = time.perf_counter()
    if K == 1:
        sc.scan(frames)
    else:
        for r in sc.scan_many((frames for _ in range(K)), count=K):
            pass
    dt = time.perf_counter() - t0
    print('K=%d: %.2f ms per clip;' % (K, dt * 1e3 / K),
          ', '.join('%s %.2f' % (k, v * 1e3 / K) for k, v in tot.items()))
