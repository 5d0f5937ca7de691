"""How many frames of pre-roll a time segment's MOG model needs before its masks agree with the
sequential run (multi-GPU segments).  Not part of the product.
    python3 tools/probe/warmup.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch
from hawkeye_amd import synth
from hawkeye_amd.detect import BackgroundSubtractorMOG
for name, spec in (('640x480', dict(width=640, height=480, fps=60.0, balls=4, frames=1300, ball_r=6.0, sigma=3.0, drift=4.0, drift_hz=0.3, distractors=4, seed=42)),
                   ('C-like 1080p', dict(synth.CLIPS['C_1080p60_9ball_noisy'], frames=1300))):
    frames = synth.ClipRenderer(spec).render(0, spec['frames'])
    w, h = spec['width'], spec['height']
    seq = BackgroundSubtractorMOG(w, h).apply_batch(frames)
    start = 1000
    for L in (0, 50, 100, 200, 300, 400, 600, 800, 1000):
        seg = BackgroundSubtractorMOG(w, h).apply_batch(frames[start - L:])[L:]
        ref = seq[start:]
        agree = (seg == ref).float().mean().item()
        fg = (ref == 255).float().mean().item()
        fg_agree = ((seg == 255) & (ref == 255)).float().sum().item() / max(1.0, (ref == 255).float().sum().item())
        first50 = (seg[:50] == ref[:50]).float().mean().item()
        print(f'{name} L={L:4d}: agreement {agree:.6f} (first 50 frames {first50:.6f}); fg fraction {fg:.5f}, fg recall {fg_agree:.4f}')
