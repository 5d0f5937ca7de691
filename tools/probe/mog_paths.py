"""Which update path the wave-frames of the MOG batch kernel take on the
benchmark clip (HK_MOG_PATH_STATS build, tools/probe/libhk_mogpaths.so)."""
import ctypes as C
import os
import sys

os.environ['HK_NATIVE_LIB'] = os.path.join(os.path.dirname(
    os.path.abspath(__file__)), 'libhk_mogpaths.so')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import _native as nat                  # noqa: E402
from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.detect import (BackgroundSubtractorMOG,  # noqa: E402
                                SimpleBlobDetector)

scene = sys.argv[1] if len(sys.argv) > 1 else 'clip'
spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
static = scene == 'static'
if static:
    spec.update(drift=0.0, drift_hz=0.0, distractors=0)
r = synth.ClipRenderer(spec, juggler=not static)
if static:
    r.discs.zero_()
frames = r.render(0, 1792)
mog = BackgroundSubtractorMOG(spec['width'], spec['height'])
blob = SimpleBlobDetector(spec['width'], spec['height'], max_frames=128)
tiles = blob.tiles_buffer()
out = (C.c_ulonglong * 8)()
nat.lib.hk_mog_path_stats.argtypes = [C.POINTER(C.c_ulonglong)]
nat.lib.hk_mog_path_stats(out)
names = ['unit', 'all first, 1 component', 'all first, 2', 'all first, 3+',
         'masked update (some lanes)', 'general step']
for s in range(0, 1792, 128):
    mog.apply_batch_tiles(frames[s:s + 128], tiles)
    nat.check(nat.lib.hk_mog_path_stats(out))
    tot = float(sum(out[:6])) or 1.0
    print('launch %2d: ' % (s // 128) + '  '.join(
        '%s %.3f' % (n, out[i] / tot) for i, n in enumerate(names)))
