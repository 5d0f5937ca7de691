#!/usr/bin/env python3
"""Per-clip stage times along a pipelined batch (ClipScanner.scan_many on the
benchmark clip).  Not part of the product.
    python3 tools/probe/batch_stages.py [clips]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402
if os.environ.get('PROBE_IMPORT_DIST'):
    import torch.distributed                            # noqa: E402,F401

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
for _ in range(5):
    sc.scan(frames)
if os.environ.get('PROBE_SINGLES'):
    for _ in range(3):
        torch.cuda.synchronize()
        sc.scan(frames)
if os.environ.get('PROBE_SET_DEVICE'):
    torch.cuda.set_device(0)
if os.environ.get('PROBE_MOG_TIMING'):
    sc.detector.mog_timing = []
for _ in range(int(os.environ.get('PROBE_WARM_BATCHES', '0'))):
    for r in sc.scan_many((frames for _ in range(n)), count=n):
        pass
torch.cuda.synchronize()
t0 = time.perf_counter()
rows = []
for r in sc.scan_many((frames for _ in range(n)), count=n):
    rows.append((time.perf_counter() - t0, dict(r.stage_seconds)))
total = time.perf_counter() - t0
print('{} clips in {:.1f} ms = {:.0f} frames/s'.format(
    n, total * 1e3, n * spec['frames'] / total))
import statistics
print('median detect {:.2f} ms'.format(statistics.median(s['detect'] for _, s in rows) * 1e3))
for k, (t, s) in enumerate(rows if os.environ.get('PROBE_ROWS') else []):
    print('clip {:2d} delivered at {:7.1f} ms  detect {:5.1f}  step3 {:6.1f} (cpu {:5.1f})  '
          'step4 {:5.1f} (cpu {:4.1f})'.format(k, t * 1e3, s['detect'] * 1e3,
                                 s['step3'] * 1e3, s['step3_cpu'] * 1e3,
                                 s['step4'] * 1e3, s['step4_cpu'] * 1e3))
