"""Scratch: run step 2 on a synthetic clip on the GPU and save the events
(frame, x, y, size) so that steps 3-4 can be profiled on the CPU."""
import sys
import numpy as np
import torch
from hawkeye_amd import synth
from hawkeye_amd.detect import EventDetector

name = sys.argv[1] if len(sys.argv) > 1 else 'C_1080p60_9ball_noisy'
spec = synth.CLIPS[name]
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
det = EventDetector(spec['width'], spec['height'], chunk=128)
out = []
for s in range(0, frames.shape[0], 128):
    out.extend(det.submit(frames[s:s + 128]))
out.extend(det.flush())
counts = np.array([len(k) for k in out])
frame = np.repeat(np.arange(len(out), dtype=np.int32), counts)
kp = np.concatenate(out).astype(np.float64)
np.savez_compressed('gpurun_out/events_{}.npz'.format(name), frame=frame,
                    x=kp[:, 0], y=kp[:, 1], r=kp[:, 2])
print(name, len(frame), 'events')
