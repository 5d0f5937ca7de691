#!/usr/bin/env python3
"""
Benchmark of the hot path: step 2 detection (MOG + blob detector) and steps
3-4 (preliminary arcs + EM) on synthetic rendered juggling video.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one whole clip through detect -> build_initial_arcs ->
EM_optimize, frames already resident in HBM (rendered on the device before
the timed region).  Workload = BASELINE.json configs[2], the 1080p60 9-ball
noisy clip the metric is quoted on (1800 frames).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank runs one such clip (own
seed) -- the clip-sharded layout of configs[4] -- and the per-rank arc tables
are gathered to rank 0 over RCCL; value = all frames of all ranks / max time.

Timed through the C ABI (events and arcs as SoA arrays); building the Python
Balltag/Ballarc objects of the notes dictionary is reported separately
(`python_notes_ms`) and is not part of `value`.

The JSON line also carries
  roofline      for the dominant kernel (mog_batch2_k), measured live with
                events on the launch stream; algorithmic bytes = 324 B per
                pixel per frame.  The kernel keeps each
                pixel's mixture in registers across the frames of a launch,
                so it moves far fewer bytes than that and `frac` can exceed 1:
                see DESIGN.md.  `traffic` comes from profiles/ (rocprofv3 PMC
                passes, collected separately) or is null.
  cpu_baseline  the CPU oracle (C port of the OpenCV algorithms, single
                thread, + the Python port of steps 3-4) timed on a bounded
                sample of the same clip on this box's host cores (rank 0, N=1).
  parity        GPU events vs the oracle's on that sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

from hawkeye_amd import synth                   # noqa: E402
from hawkeye_amd.arcs import Clip, scan_params  # noqa: E402
from hawkeye_amd.detect import EventDetector    # noqa: E402

WORKLOAD = 'C_1080p60_9ball_noisy'
CHUNK = 128
ALGO_BYTES_PER_PIXEL_FRAME = 324       # 3 in + 1 out + 160 model rd + 160 wr
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8.0 TB/s spec


def events_to_soa(per_frame, spec):
    """Keypoints -> the SoA arrays of notes['meas'] (same float64 expressions
    as reference tracker.py:335-342, 413-417 with scan video == source)."""
    counts = np.fromiter((len(k) for k in per_frame), np.int64,
                         len(per_frame))
    frame = np.repeat(np.arange(len(per_frame), dtype=np.int32), counts)
    kp = (np.concatenate(per_frame) if counts.sum() else
          np.zeros((0, 3), np.float32)).astype(np.float64)
    h, w = spec['height'], spec['width']
    scale = h / h
    pad_left = (w - h * (w / h)) / 2
    return frame, pad_left + kp[:, 0] * scale, kp[:, 1] * scale, \
        kp[:, 2] * scale


class Pipeline:
    def __init__(self, spec, frames, device):
        self.spec, self.frames, self.device = spec, frames, device
        self.det = EventDetector(spec['width'], spec['height'], chunk=CHUNK,
                                 device=device)
        self.params = scan_params(spec['height'], spec['height'], spec['fps'])
        self.mog_ms = []            # per-launch durations of the MOG kernel
        self.mog_frames = []
        self.time_mog = False
        self.stage_s = {'detect': 0.0, 'step3': 0.0, 'step4': 0.0}

    def detect(self):
        det, frames = self.det, self.frames
        det.mog.reset()
        out = []
        pairs = []
        orig = det.mog.apply_batch
        if self.time_mog:
            def timed(fr, out=None):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig(fr, out=out)
                e1.record()
                pairs.append((e0, e1, fr.shape[0]))
                return r
            det.mog.apply_batch = timed
        try:
            for s in range(0, frames.shape[0], CHUNK):
                out.extend(det.submit(frames[s:s + CHUNK]))
            out.extend(det.flush())
        finally:
            det.mog.apply_batch = orig
        for e0, e1, n in pairs:
            self.mog_ms.append(e0.elapsed_time(e1))
            self.mog_frames.append(n)
        return out

    def step(self):
        spec = self.spec
        t0 = time.perf_counter()
        per_frame = self.detect()
        frame, x, y, r = events_to_soa(per_frame, spec)
        t1 = time.perf_counter()
        clip = Clip(spec['width'], spec['height'], spec['fps'],
                    self.frames.shape[0], frame, x, y, r, params=self.params,
                    device=self.device)
        clip.build_initial_arcs()
        t2 = time.perf_counter()
        clip.em_optimize()
        params, ids, members = clip.arcs()
        alive, final = clip.tag_state()
        tilt = clip.state()['camera_tilt']
        clip.close()
        t3 = time.perf_counter()
        self.stage_s['detect'] += t1 - t0
        self.stage_s['step3'] += t2 - t1
        self.stage_s['step4'] += t3 - t2
        return per_frame, params, int(alive.sum()), tilt


def cpu_baseline(spec, frames_dev, gpu_events, sample):
    """Oracle on the first `sample` frames: C MOG + SimpleBlobDetector (one
    thread), then the Python steps 3-4.  Also the parity check."""
    from oracle import detect_ref as D
    from oracle import em_ref
    host = frames_dev[:sample].cpu().numpy()
    mog = D.Mog(spec['width'], spec['height'])
    bp = D.hawkeye_params(spec['height'])
    t0 = time.perf_counter()
    events = [D.blob_detect(mog.apply(f), bp) for f in host]
    t1 = time.perf_counter()
    meas = {f: [(float(k[0]), float(k[1]), float(k[2])) for k in ev]
            for f, ev in enumerate(events)}
    sc = em_ref.Scene(meas, spec['width'], spec['height'], spec['fps'],
                      frame_count=sample)
    sc.build_initial_arcs()
    sc.em_optimize()
    t2 = time.perf_counter()
    exact = all(np.array_equal(np.asarray(a).view(np.uint32),
                               np.asarray(b).view(np.uint32))
                for a, b in zip(gpu_events[:sample], events))
    base = {
        'value': round(sample / (t2 - t0), 3), 'unit': 'frames/s',
        'cores': 1, 'kind': 'port',
        'sample': 'first {} frames of the same clip: C oracle MOG + '
                  'SimpleBlobDetector {:.2f} s, Python oracle steps 3-4 '
                  '{:.2f} s, single thread'.format(sample, t1 - t0, t2 - t1),
        'detect_only_fps': round(sample / (t1 - t0), 3),
    }
    parity = {'events_bit_exact_vs_oracle': bool(exact),
              'frames_compared': sample,
              'events_compared': int(sum(len(e) for e in events))}
    return base, parity


def traffic_from_profiles():
    """Per-launch HBM bytes of mog_batch2_k from the committed PMC summary
    (profiles/mog_traffic.json, written by tools/pmc_traffic.py), or None."""
    path = os.path.join(REPO, 'profiles', 'mog_traffic.json')
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=None,
                    help='shorter clip (debugging only; invalidates value)')
    ap.add_argument('--cpu-sample', type=int, default=96)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus {} must be launched with '
                     'torch.distributed.run --nproc-per-node {}'.format(
                         args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))

    spec = dict(synth.CLIPS[WORKLOAD])
    spec['seed'] += rank                    # one clip per GPU
    if args.frames:
        spec['frames'] = args.frames
    renderer = synth.ClipRenderer(spec, device=local)
    frames = renderer.render(0, spec['frames'])
    torch.cuda.synchronize()
    pipe = Pipeline(spec, frames, local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        result = pipe.step()
    pipe.stage_s = {k: 0.0 for k in pipe.stage_s}
    pipe.time_mog = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = pipe.step()
    barrier()
    elapsed = time.perf_counter() - t0

    per_frame, arc_params, tags_left, tilt = result
    # collect every rank's arc table on rank 0 (RCCL), and the slowest time
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n_arcs = torch.tensor([len(arc_params)], dtype=torch.int64,
                              device='cuda')
        counts = [torch.zeros_like(n_arcs) for _ in range(world)]
        dist.all_gather(counts, n_arcs)
        cap = int(max(c.item() for c in counts))
        mine = torch.zeros((cap, 5), dtype=torch.float64, device='cuda')
        mine[:len(arc_params)] = torch.from_numpy(arc_params).cuda()
        tables = [torch.zeros_like(mine) for _ in range(world)] \
            if rank == 0 else None
        dist.gather(mine, tables, dst=0)
        total_arcs = int(sum(c.item() for c in counts))
    else:
        total_arcs = len(arc_params)

    if rank != 0:
        dist.destroy_process_group()
        return

    total_frames = spec['frames'] * args.steps * world
    value = total_frames / elapsed
    px = spec['width'] * spec['height']
    mog_ms = np.array(pipe.mog_ms)
    mog_fr = np.array(pipe.mog_frames)
    full = mog_fr == CHUNK                  # launches of a whole chunk
    launch_ms = float(mog_ms[full].mean()) if full.any() else \
        float(mog_ms.mean())
    launch_frames = CHUNK if full.any() else float(mog_fr.mean())
    achieved = ALGO_BYTES_PER_PIXEL_FRAME * px * launch_frames / \
        (launch_ms * 1e-3) / 1e9
    traffic = traffic_from_profiles()
    roofline = {
        'bound': 'hbm', 'kernel': 'mog_batch2_k',
        'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 4),
        'traffic': None if traffic is None else traffic.get('bytes_per_launch'),
        'launch_ms': round(launch_ms, 4),
        'frames_per_launch': int(launch_frames),
        'algorithmic_bytes_per_launch': int(ALGO_BYTES_PER_PIXEL_FRAME * px *
                                            launch_frames),
        'kernel_fps': round(launch_frames / (launch_ms * 1e-3), 1),
        'note': 'model stays in registers across the frames of a launch, so '
                'physical traffic is ~6.5 B/px/frame, not 324: frac > 1 is '
                'real, the kernel is VALU-bound (DESIGN.md)',
    }
    if traffic is not None:
        roofline['traffic_source'] = traffic.get('source')

    out = {
        'metric': '1080p frames/sec (detect+EM)',
        'value': round(value, 1), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (MOG) / int64+f64 (blob) / f64 (EM)',
        'data': 'synthetic',
        'config': {'workload': WORKLOAD, 'width': spec['width'],
                   'height': spec['height'], 'fps': spec['fps'],
                   'frames_per_clip': spec['frames'],
                   'clips': world, 'sharding': 'one clip per GPU, arc tables '
                   'gathered to rank 0 over RCCL' if world > 1 else 'single GPU',
                   'detect_chunk_frames': CHUNK},
        'stage_ms_per_step': {k: round(v / args.steps * 1e3, 3)
                              for k, v in pipe.stage_s.items()},
        'result': {'events': int(sum(len(k) for k in per_frame)),
                   'arcs': total_arcs, 'tags_on_arcs': tags_left,
                   'camera_tilt_rad': tilt},
        'roofline': roofline,
    }

    if world == 1 and not args.no_cpu_baseline:
        sample = min(args.cpu_sample, spec['frames'])
        base, parity = cpu_baseline(spec, frames, per_frame, sample)
        out['cpu_baseline'] = base
        out['parity'] = parity
        out['speedup_vs_cpu_baseline'] = round(value / base['value'], 1)

    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
