"""Scratch probe: when each clip of a batch leaves each stage of scan_many."""
import sys, time
import torch
sys.path.insert(0, '.')
from hawkeye_amd import synth
from hawkeye_amd.batch import ClipScanner

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec, device=0).render(0, spec['frames'])
torch.cuda.synchronize()
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
for _ in range(3):
    sc.scan(frames)
log = []
orig_build, orig_refine, orig_detect = sc.build_arcs, sc.refine_arcs, sc.detect


def detect(f):
    r = orig_detect(f)
    log.append(('detected', time.perf_counter()))
    return r


def build(*a, **k):
    t = time.perf_counter()
    r = orig_build(*a, **k)
    log.append(('built', t, time.perf_counter()))
    return r


def refine(*a, **k):
    t = time.perf_counter()
    r = orig_refine(*a, **k)
    log.append(('refined', t, time.perf_counter()))
    return r


sc.detect, sc.build_arcs, sc.refine_arcs = detect, build, refine
t0 = time.perf_counter()
n = 0
for r in sc.scan_many((frames for _ in range(K)), count=K):
    n += 1
t1 = time.perf_counter()
print('K=%d total %.1f ms (%.2f per clip)' % (K, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / K))
det = [e[1] for e in log if e[0] == 'detected']
print('detect done at', ' '.join('%.0f' % ((t - t0) * 1e3) for t in det))
b = sorted((e[1], e[2]) for e in log if e[0] == 'built')
print('build  start-end', ' '.join('%.0f-%.0f' % ((a - t0) * 1e3, (c - t0) * 1e3) for a, c in b))
r = sorted((e[1], e[2]) for e in log if e[0] == 'refined')
print('refine start-end', ' '.join('%.0f-%.0f' % ((a - t0) * 1e3, (c - t0) * 1e3) for a, c in r))
