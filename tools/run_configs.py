#!/usr/bin/env python3
"""Runs the BASELINE.json configurations that are not the bench workload on
ONE GPU and prints a JSON line per configuration (for DESIGN.md §5):

  B  720p30 5-ball clip, steps 2-4                       (configs[1])
  D  4K 240 fps 9-ball long clip, 14 400 frames          (configs[3]):
     the sequential scan, and the 2/4/8-way time-segmented layout emulated
     on this GPU segment after segment (same arithmetic as one rank per GPU;
     the time is the slowest segment, i.e. what G GPUs would take for step 2)
  E  64 clips 1080p60 x 600 frames, pipelined batch      (configs[4])

    python tools/run_configs.py [B] [D] [E]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from hawkeye_amd import dist as hkdist, synth               # noqa: E402
from hawkeye_amd.batch import ClipScanner                   # noqa: E402
from hawkeye_amd.detect import EventDetector                # noqa: E402
from hawkeye_amd.frames import SyntheticSource              # noqa: E402


def summary(res):
    return {'events': int(len(res.frame)), 'arcs': int(len(res.arc_params)),
            'tags_on_arcs': int(res.alive.sum()),
            'camera_tilt_rad': res.camera_tilt,
            'g_px_per_frame_sq': res.g_px_per_frame_sq}


def run_B():
    spec = synth.CLIPS['B_720p30_5ball']
    frames = synth.ClipRenderer(spec).render(0, spec['frames'])
    torch.cuda.synchronize()
    sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
    sc.scan(frames)
    t0 = time.perf_counter()
    res = sc.scan(frames)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = 0
    for _ in sc.scan_many(frames for _ in range(10)):
        n += spec['frames']
    piped = n / (time.perf_counter() - t0)
    sc.close()
    return {'config': 'B_720p30_5ball', 'frames': spec['frames'],
            'single_clip_ms': round(dt * 1e3, 2),
            'single_clip_fps': round(spec['frames'] / dt, 1),
            'pipelined_fps': round(piped, 1),
            'stage_ms': {k: round(v * 1e3, 2)
                         for k, v in res.stage_seconds.items()},
            'result': summary(res)}


def run_D():
    spec = synth.CLIPS['D_4k240_9ball_long']
    w, h, F = spec['width'], spec['height'], spec['frames']
    out = {'config': 'D_4k240_9ball_long', 'frames': F}
    sc = ClipScanner(w, h, spec['fps'])
    t0 = time.perf_counter()
    res = sc.scan(SyntheticSource(spec))
    dt = time.perf_counter() - t0
    out['sequential'] = {
        'seconds_incl_rendering': round(dt, 3),
        'fps_incl_rendering': round(F / dt, 1),
        'stage_ms': {k: round(v * 1e3, 1)
                     for k, v in res.stage_seconds.items()},
        'result': summary(res)}
    whole = res.events
    for G in (2, 4, 8):
        worst = 0.0
        parts = []
        for first, keep, end in hkdist.segment_bounds(F, G, 400):
            det = EventDetector(w, h)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            parts += hkdist.detect_segment(SyntheticSource(spec), det, first,
                                           keep, end)
            worst = max(worst, time.perf_counter() - t0)
            det.close()
        assert len(parts) == F
        matched = total = 0
        for a, b in zip(parts, whole):
            total += len(b)
            if len(a) and len(b):
                d = np.hypot(b[:, None, 0] - a[None, :, 0],
                             b[:, None, 1] - a[None, :, 1])
                matched += int((d.min(axis=1) <= 0.5).sum())
        r = sc.refine_arcs(*sc.build_arcs(parts))
        out['segmented_{}'.format(G)] = {
            'slowest_segment_seconds_incl_rendering': round(worst, 3),
            'step2_fps_if_one_gpu_per_segment': round(F / worst, 1),
            'sequential_events_matched_within_half_px':
                round(matched / max(1, total), 5),
            'events': int(sum(len(p) for p in parts)),
            'arcs': int(len(r.arc_params)),
            'arcs_sequential': int(len(res.arc_params))}
    sc.close()
    return out


def run_E():
    specs = [synth.clip_E(i) for i in range(64)]
    sc = ClipScanner(1920, 1080, 60.0)

    def clips():
        for s in specs:
            yield synth.ClipRenderer(s).render(0, s['frames'])
    list(sc.scan_many(synth.ClipRenderer(specs[i]).render(0, 600)
                      for i in range(2)))          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = list(sc.scan_many(clips()))
    dt = time.perf_counter() - t0
    sc.close()
    arcs = [len(r.arc_params) for r in results]
    return {'config': 'E_64x1080p60x600', 'clips': len(results),
            'frames': 64 * 600, 'seconds_incl_rendering': round(dt, 3),
            'fps_incl_rendering': round(64 * 600 / dt, 1),
            'arcs_per_clip_min_max': [min(arcs), max(arcs)],
            'events_total': int(sum(len(r.frame) for r in results))}


if __name__ == '__main__':
    want = [a.upper() for a in sys.argv[1:]] or ['B', 'D', 'E']
    for name in want:
        print(json.dumps({'B': run_B, 'D': run_D, 'E': run_E}[name]()),
              flush=True)
