#!/usr/bin/env python3
"""The blob kernels alone on the GPU, for rocprofv3 passes and timing (not
part of the product): MOG runs over the benchmark clip up to chunk --chunk-no,
whose 1-bit tiles are then detected --reps times with nothing else running.

    python3 tools/blob_prof.py [--chunk-no 7] [--reps 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.detect import (BackgroundSubtractorMOG,  # noqa: E402
                                SimpleBlobDetector)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunk-no', type=int, default=7)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--chunk', type=int, default=128)
    args = ap.parse_args()
    spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
    n = (args.chunk_no + 1) * args.chunk
    frames = synth.ClipRenderer(spec).render(0, n)
    w, h = spec['width'], spec['height']
    mog = BackgroundSubtractorMOG(w, h)
    blob = SimpleBlobDetector(w, h, max_frames=args.chunk)
    tiles = blob.tiles_buffer()
    for s in range(0, n, args.chunk):
        mog.apply_batch_tiles(frames[s:s + args.chunk], tiles)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        blob.enqueue_tiles(args.chunk)
        kps = blob.finish()
    dt = (time.perf_counter() - t0) / args.reps
    print('chunk {}: {} keypoints, {:.3f} ms per detection of {} frames '
          '(host wall, kernels + read-back + grouping)'.format(
              args.chunk_no, sum(len(k) for k in kps), dt * 1e3, args.chunk))


if __name__ == '__main__':
    main()
