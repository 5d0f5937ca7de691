#!/usr/bin/env python3
"""How much of a clip's detection time is the MOG chain and how much the blob
kernels add beside it: the clip's MOG launches back to back with and without
the blob chain of every chunk on the side stream.  Not part of the product."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import synth                           # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
det = sc.detector
for _ in range(3):
    sc.detect(frames)


def timed(fn, reps=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def mog_only():
    det.mog.reset()
    for s in range(0, frames.shape[0], 128):
        det.mog.apply_batch_tiles(frames[s:s + 128], det._tiles[0])


side = torch.cuda.Stream()


def mog_and_blob_no_host():
    """MOG chain + blob chains on the side stream, results never read back
    per chunk (one finish at the end of every third chunk's handle)."""
    det.mog.reset()
    k = 0
    for s in range(0, frames.shape[0], 128):
        p = k % 3
        if k >= 3:
            det.blob[p].finish()
        n = min(128, frames.shape[0] - s)
        det.mog.apply_batch_tiles(frames[s:s + n], det._tiles[p])
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        det.blob[p].enqueue_tiles(n, stream=side)
        k += 1
    for p in range(3):
        det.blob[(k + p) % 3].finish()


print('MOG chain alone          {:.2f} ms per clip'.format(timed(mog_only)))
print('MOG + blob, depth 3      {:.2f} ms per clip'.format(
    timed(mog_and_blob_no_host)))
print('ClipScanner.detect       {:.2f} ms per clip'.format(
    timed(lambda: sc.detect(frames))))

# the whole clip's blob detection in one batch after the MOG chain
from hawkeye_amd.detect import SimpleBlobDetector, hawkeye_blob_params  # noqa: E402
big = SimpleBlobDetector(spec['width'], spec['height'],
                         hawkeye_blob_params(spec['height']),
                         max_frames=frames.shape[0])
big_tiles = big.tiles_buffer()
fb = (spec['width'] // 32) * (spec['height'] // 4) * 16


def mog_then_blob():
    det.mog.reset()
    for s in range(0, frames.shape[0], 128):
        det.mog.apply_batch_tiles(frames[s:s + 128], big_tiles + s * fb)
    big.enqueue_tiles(frames.shape[0])
    return big.finish()


def blob_only():
    big.enqueue_tiles(frames.shape[0])
    return big.finish()


print('MOG chain, then one blob batch {:.2f} ms per clip'.format(
    timed(mog_then_blob)))
print('one blob batch of the clip     {:.2f} ms'.format(timed(blob_only)))
ref = sc.detect(frames)
got = mog_then_blob()
import numpy as np                                      # noqa: E402
print('same events:', len(ref) == len(got) and all(
    np.array_equal(a, b) for a, b in zip(ref, got)))

# host share of finish(): GPU work complete before the call
b0 = det.blob[0]
ts = []
for s in range(256, 256 + 5 * 128, 128):
    det.mog.apply_batch_tiles(frames[s:s + 128], det._tiles[0])
    b0.enqueue_tiles(128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = b0.finish()
    ts.append((time.perf_counter() - t0) * 1e6)
print('finish() after the GPU is done (host only), us per 128 frames:',
      ' '.join('%.0f' % t for t in ts), 'keypoints', sum(len(x) for x in k))
big.enqueue_tiles(frames.shape[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
big.finish()
print('same for the whole clip: %.0f us' % ((time.perf_counter() - t0) * 1e6))
