"""When the waves of a MOG batch launch start and end (HK_MOG_WAVE_TIMES
build, tools/probe/libhk_mogtimes.so; tools/probe/build_probe_libs.sh): is a
launch bound by its slowest waves or by throughput?

    python3 tools/probe/mog_timeline.py [clip|static] [launch index ...]
"""
import ctypes as C
import os
import sys

import numpy as np

os.environ['HK_NATIVE_LIB'] = os.path.join(os.path.dirname(
    os.path.abspath(__file__)), 'libhk_mogtimes.so')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.detect import (BackgroundSubtractorMOG,  # noqa: E402
                                SimpleBlobDetector)

scene = sys.argv[1] if len(sys.argv) > 1 else 'clip'
which = [int(a) for a in sys.argv[2:]] or [2, 7, 14]
CHUNK = int(os.environ.get("PROBE_CHUNK", "120"))
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
static = scene == 'static'
if static:
    spec.update(drift=0.0, drift_hz=0.0, distractors=0)
r = synth.ClipRenderer(spec, juggler=not static)
if static:
    r.discs.zero_()
frames = r.render(0, 1800)
w, h = spec['width'], spec['height']
mog = BackgroundSubtractorMOG(w, h)
blob = SimpleBlobDetector(w, h, max_frames=CHUNK)
tiles = blob.tiles_buffer()
n_groups = (w * h // 2 + 255) // 256
blocks = (n_groups + 7) // 8 * 8 * 4
rec = torch.zeros((blocks, 4), dtype=torch.int64, device='cuda')
nat.lib.hk_mog_wave_times.argtypes = [C.c_void_p]
for k, s in enumerate(range(0, 1800, CHUNK)):
    on = k in which
    nat.check(nat.lib.hk_mog_wave_times(rec.data_ptr() if on else None))
    rec.zero_()
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    mog.apply_batch_tiles(frames[s:s + CHUNK], tiles)
    e1.record()
    torch.cuda.synchronize()
    if not on:
        continue
    a = rec.cpu().numpy()
    a = a[a[:, 1] > 0]
    t0, t1 = a[:, 0] * 10e-3, a[:, 1] * 10e-3       # us
    slow = a[:, 2] & 0xffffffff
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    dur = t1 - t0
    span = t1.max()
    print('launch %d (%s): %.1f us by events, waves %d, span of wave times '
          '%.1f us' % (k, scene, e0.elapsed_time(e1) * 1e3, len(a), span))
    print('  wave duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max '
          '%.1f;  sum %.0f us = %.2f of (span x 4096 slots)' % (
              dur.mean(), np.median(dur), np.percentile(dur, 90),
              np.percentile(dur, 99), dur.max(), dur.sum(),
              dur.sum() / (span * 4096)))
    # waves in flight over time
    edges = np.linspace(0, span, 21)
    live = [(np.minimum(t1, hi) - np.maximum(t0, lo)).clip(0).sum() / (hi - lo)
            for lo, hi in zip(edges[:-1], edges[1:])]
    print('  waves in flight per 5 % of the span: ' +
          ' '.join('%d' % x for x in live))
    # by class of slow frames
    for lo, hi in ((0, 0), (1, 10), (11, 40), (41, 90), (91, 10**6)):
        m = (slow >= lo) & (slow <= hi)
        if m.any():
            print('  slow frames %3d-%-3s: %5d waves, duration mean %.1f max '
                  '%.1f us, start mean %.1f, end max %.1f' % (
                      lo, hi if hi < 10**6 else '', m.sum(), dur[m].mean(),
                      dur[m].max(), t0[m].mean(), t1[m].max()))
    last = np.argsort(t1)[-8:]
    print('  last waves to end: ' + '  '.join(
        '(start %.0f dur %.0f slow %d)' % (t0[i], dur[i], slow[i])
        for i in last))
