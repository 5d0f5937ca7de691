#!/usr/bin/env python3
"""MOG batch launches alone on the GPU, for rocprofv3 counter passes and
timing (not part of the product).

    python3 tools/mog_prof.py [--scene clip|static] [--frames 768] [--reps 2]

scene clip   = bench.py's workload (C_1080p60_9ball_noisy), first --frames
scene static = the same background and sensor noise without balls,
               distractors, juggler or drift: every pixel stays on its first
               mixture component (the kernel's packed fast step only)
Launches of 128 frames through hk_mog_apply_batch_tiles, the entry point the
detection pipeline uses; prints the average launch time (HIP events).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.detect import (BackgroundSubtractorMOG,  # noqa: E402
                                SimpleBlobDetector)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scene', default='clip')
    ap.add_argument('--frames', type=int, default=768)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--chunk', type=int, default=128)
    ap.add_argument('--config', default='C_1080p60_9ball_noisy')
    ap.add_argument('--masks', action='store_true',
                    help='byte masks (hk_mog_apply_batch) instead of tiles')
    ap.add_argument('--no-frame-traffic', action='store_true',
                    help='probe library only (HK_NATIVE_LIB=tools/probe/'
                         'libhk_mogtimes.so): frame loads dropped by the '
                         'buffer range check, all instructions kept')
    ap.add_argument('--no-reset', action='store_true',
                    help='the model carries on from repetition to repetition')
    args = ap.parse_args()
    spec = dict(synth.CLIPS[args.config])
    spec['frames'] = max(args.frames, 8)
    static = args.scene == 'static'
    if static:
        spec.update(drift=0.0, drift_hz=0.0, distractors=0)
    r = synth.ClipRenderer(spec, juggler=not static)
    if static:
        r.discs.zero_()
    frames = r.render(0, args.frames)
    w, h = spec['width'], spec['height']
    mog = BackgroundSubtractorMOG(w, h)
    blob = SimpleBlobDetector(w, h, max_frames=args.chunk)
    tiles = blob.tiles_buffer()
    masks = torch.empty((args.chunk, h, w), dtype=torch.uint8,
                        device='cuda') if args.masks else None
    torch.cuda.synchronize()
    if args.no_frame_traffic:
        from hawkeye_amd import _native as nat
        nat.check(nat.lib.hk_mog_frame_records(0))
    times = []
    for rep in range(args.reps):
        if not args.no_reset or rep == 0:
            mog.reset()
        for s in range(0, args.frames - args.chunk + 1, args.chunk):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if args.masks:
                mog.apply_batch(frames[s:s + args.chunk], out=masks)
            else:
                mog.apply_batch_tiles(frames[s:s + args.chunk], tiles)
            e1.record()
            times.append((rep, s, e0, e1))
    torch.cuda.synchronize()
    ms = [(rep, s, e0.elapsed_time(e1)) for rep, s, e0, e1 in times]
    later = [t for rep, s, t in ms if s >= 2 * args.chunk] or \
        [t for _, _, t in ms]
    print('scene {} {}x{} chunk {}: launches {} avg {:.4f} ms (from the third '
          'chunk on: {:.4f} ms, min {:.4f}, max {:.4f})'.format(
              args.scene, w, h, args.chunk, len(ms),
              sum(t for _, _, t in ms) / len(ms), sum(later) / len(later),
              min(later), max(later)))
    print('per launch ms:', ' '.join('{:.3f}'.format(t) for _, _, t in ms))


if __name__ == '__main__':
    main()
