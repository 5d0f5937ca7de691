#!/usr/bin/env python3
"""Step 2 on the 64 clips of configuration E through both hand-over paths
(1-bit tiles, byte masks); where they differ, or on --oracle N clips anyway,
the frames are checked against the C oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import synth                                   # noqa: E402
from hawkeye_amd.detect import EventDetector, hawkeye_blob_params   # noqa: E402
from oracle import detect_ref as D                              # noqa: E402


def run(frames, spec, tiles):
    os.environ['HK_DETECT_TILES'] = '1' if tiles else '0'
    det = EventDetector(spec['width'], spec['height'],
                        hawkeye_blob_params(spec['height']), chunk=128)
    out = []
    for s in range(0, frames.shape[0], 128):
        out += det.submit(frames[s:s + 128])
    out += det.flush()
    det.close()
    return out


def main():
    clips = [int(a) for a in sys.argv[1:]] or range(64)
    bad = 0
    for i in clips:
        spec = synth.clip_E(i)
        frames = synth.ClipRenderer(spec).render(0, spec['frames'])
        a = run(frames, spec, True)
        b = run(frames, spec, False)
        same = all(np.array_equal(x, y) for x, y in zip(a, b))
        na, nb = sum(len(x) for x in a), sum(len(x) for x in b)
        print('clip', i, 'tiles', na, 'bytes', nb, 'same' if same else 'DIFFER',
              flush=True)
        if same and len(sys.argv) < 2:
            continue
        host = frames.cpu().numpy()
        mog = D.Mog(spec['width'], spec['height'])
        bp = D.hawkeye_params(spec['height'])
        for f in range(spec['frames']):
            want = D.blob_detect(mog.apply(host[f]), bp)
            ta = np.array_equal(want.view(np.uint32), a[f].view(np.uint32))
            tb = np.array_equal(want.view(np.uint32), b[f].view(np.uint32))
            if not (ta and tb):
                print('  frame', f, 'oracle', len(want), 'tiles', len(a[f]),
                      'ok' if ta else 'WRONG', 'bytes', len(b[f]),
                      'ok' if tb else 'WRONG', flush=True)
                bad += 1
                if bad > 6:
                    return 1
    return 1 if bad else 0


if __name__ == '__main__':
    torch.cuda.set_device(0)
    sys.exit(main())
