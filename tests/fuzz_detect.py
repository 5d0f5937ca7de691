#!/usr/bin/env python3
"""Randomised step-2 parity run on the GPU: random clip geometries (odd sizes
included), noise levels, distractor counts and chunk lengths through
EventDetector, every frame's keypoints and the final MOG model compared bit
for bit with the C oracle (oracle/detect_ref).  Stops at the first mismatch.

    python tests/fuzz_detect.py [--seconds 120] [--seed 1] [--max-pixels 300000]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import synth                               # noqa: E402
from hawkeye_amd.detect import EventDetector, hawkeye_blob_params   # noqa: E402
from oracle import detect_ref as D                          # noqa: E402


def random_spec(rng, max_pixels):
    while True:
        w = int(rng.choice([rng.randint(40, 200), rng.randint(200, 1000),
                            64, 128, 640, 641, 639]))
        h = int(rng.choice([rng.randint(30, 160), rng.randint(160, 600),
                            48, 480, 479, 481]))
        if rng.uniform() < 0.5:     # frames made of 32x4 tiles: the MOG
            w = max(32, w // 32 * 32)       # kernel hands 1-bit tiles over
            h = max(4, h // 4 * 4)
        if w * h <= max_pixels:
            break
    return dict(width=w, height=h, fps=float(rng.choice([24, 30, 60, 120])),
                balls=int(rng.randint(1, 10)),
                frames=int(rng.randint(3, 80) if rng.uniform() < 0.7
                           else rng.randint(200, 260)),
                ball_r=float(rng.uniform(1.5, 14.0)),
                sigma=float(rng.choice([0.0, 1.0, 3.0, 6.0, 12.0])),
                drift=float(rng.choice([0.0, 8.0, 30.0])),
                drift_hz=float(rng.choice([0.0, 0.2, 2.0])),
                distractors=int(rng.choice([0, 5, 20])),
                seed=int(rng.randint(1, 1 << 30)))


def run_case(spec, chunk):
    frames = synth.ClipRenderer(spec).render(0, spec['frames'])
    det = EventDetector(spec['width'], spec['height'],
                        hawkeye_blob_params(spec['height']), chunk=chunk)
    got = []
    for s in range(0, spec['frames'], chunk):
        got += det.submit(frames[s:s + chunk])
    got += det.flush()
    model = det.mog.export_model()
    det.close()
    host = frames.cpu().numpy()
    mog = D.Mog(spec['width'], spec['height'])
    bp = D.hawkeye_params(spec['height'])
    n = 0
    for f, frame in enumerate(host):
        want = D.blob_detect(mog.apply(frame), bp)
        if want.shape != got[f].shape or not np.array_equal(
                want.view(np.uint32), got[f].view(np.uint32)):
            return 'frame {}: {} keypoints, oracle {}'.format(
                f, len(got[f]), len(want)), n
        n += len(want)
    want_model = np.asarray(mog.model())
    if not np.array_equal(np.asarray(model).view(np.uint32).ravel(),
                          want_model.view(np.uint32).ravel()):
        return 'MOG model differs after the last frame', n
    return None, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=120.0)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--max-pixels', type=int, default=300000)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0 = said = time.time()
    cases = events = 0
    while time.time() - t0 < args.seconds:
        spec = random_spec(rng, args.max_pixels)
        chunk = int(rng.choice([1, 2, 7, 32, 128]))
        err, n = run_case(spec, chunk)
        cases += 1
        if time.time() - said > 30:     # a silent GPU run is taken to be hung
            print('...', cases, 'cases', flush=True)
            said = time.time()
        events += n
        if err:
            print('MISMATCH', err, 'chunk', chunk, spec, flush=True)
            sys.exit(1)
        if cases % 10 == 0:
            print('{} cases, {} keypoints, {:.0f} s'.format(
                cases, events, time.time() - t0), flush=True)
    print('ok: {} cases, {} keypoints compared'.format(cases, events))


if __name__ == '__main__':
    torch.cuda.set_device(0)
    main()
