#!/usr/bin/env python3
"""
Benchmark of the hot path: step 2 detection (MOG + blob detector) and steps
3-4 (preliminary arcs + EM) on synthetic rendered juggling video.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--layout clips|segments]

With --gpus N > 1 and no launcher around it (WORLD_SIZE unset) the command
starts its N ranks itself (torch.distributed.run as a child process; the
parent touches no GPU) and relays rank 0's JSON line; under
`python -m torch.distributed.run ... bench.py --gpus N` it is a rank.

--layout clips (default).  One "step" = one whole clip through detect ->
build_initial_arcs -> EM_optimize, frames already resident in HBM (rendered on
the device before the timed region).  Workload = BASELINE.json configs[2], the
1080p60 9-ball noisy clip the metric is quoted on (1800 frames).  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank runs one such
clip (own seed) -- the clip-sharded layout of configs[4] -- and the per-rank
arc tables are gathered to rank 0 over RCCL; value = all frames of all ranks /
max time ("weak" scaling).  The K timed steps run the way a batch of clips
runs in production (hawkeye_amd.batch.ClipScanner.scan_many): the GPU detects
pass k+1, k+2 ... while the host cores run steps 3-4 of the passes before.
Every pass does all of the work, nothing is cached between passes.  The
latency of one clip on its own (no overlap) is reported as `single_clip_ms`.

--layout segments.  BASELINE.json configs[3]: ONE long clip (4K 240 fps,
14400 frames) time-segmented over the N ranks (hawkeye_amd.dist): rank r
detects frames [start_r - 400, end_r) on its GPU with a cold MOG model, the
kept events are gathered on rank 0 over RCCL, and rank 0 runs steps 3-4 on the
whole clip.  One "step" = the whole clip once; value = frames of the clip /
max-over-ranks time including the gather and rank 0's steps 3-4 ("strong"
scaling).  A rank's frames are resident in HBM before the timed region when
they fit (358 GB for the whole clip: they do from N = 2 on); at N = 1 they
are rendered chunk by chunk inside the timed region and the line says so.

Timed through the C ABI (events and arcs as SoA arrays), without building the
Python Balltag/Ballarc objects of the notes dictionary.

The JSON line also carries
  roofline      for the dominant kernel (mog_batch2_k), launch time measured
                live with events on the launch stream.  The kernel keeps each
                pixel's mixture on chip across the frames of a launch, so HBM
                is not what bounds it: the bound is VALU issue (one vector
                instruction per 4 cycles per SIMD for the packed-f32, convert,
                compare and 64-bit move instructions it is made of; measured,
                profiles/r03_ubench_counters.csv).  achieved = VALU
                instructions of a launch (SQ counters of the shipped kernel on
                this clip, profiles/mog_issue.json) / launch time; peak = 1024
                SIMDs x 2.4 GHz / 4.  The physical HBM rate (PMC traffic /
                launch time) and SURVEY.md section 8d's algorithmic stream
                rate are given beside it.
  cpu_baseline  the CPU oracle (C port of the OpenCV algorithms + the Python
                port of steps 3-4) timed on a bounded sample of the same clip
                on this box's host cores (rank 0, N=1): single thread, and
                the detection threaded over rows / frames on the rank's cores.
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

# torch and the package are imported by load_runtime(), after main() has
# decided whether this process is a rank or only the parent that starts the
# ranks: the parent of a multi-GPU run never loads them and never touches a GPU
torch = dist = hkdist = synth = ClipScanner = host_core_share = None


def load_runtime():
    global torch, dist, hkdist, synth, ClipScanner, host_core_share
    import torch as _torch
    import torch.distributed as _dist
    from hawkeye_amd import dist as _hkdist
    from hawkeye_amd import synth as _synth
    from hawkeye_amd.batch import ClipScanner as _scanner
    from hawkeye_amd.batch import host_core_share as _share
    torch, dist, hkdist, synth = _torch, _dist, _hkdist, _synth
    ClipScanner, host_core_share = _scanner, _share


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started on its own: run the N ranks as
    children (torch.distributed.run, one process per GPU, rendezvous on
    127.0.0.1), let rank 0's JSON line through on stdout and return their exit
    status.  This parent holds no GPU state (nothing here imports torch)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


WORKLOAD = 'C_1080p60_9ball_noisy'
SEGMENT_WORKLOAD = 'D_4k240_9ball_long'
CHUNK = int(os.environ.get('HK_BENCH_CHUNK', '360'))
ALGO_BYTES_PER_PIXEL_FRAME = 324       # 3 in + 1 out + 160 model rd + 160 wr
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 1024                           # 256 CUs x 4
NOMINAL_CLOCK_HZ = 2.4e9
VALU_ISSUE_CYCLES = 4                  # profiles/r03_ubench_counters.csv
# issue cost by class on a busy SIMD, same calibration: packed f32, convert,
# compare and 64-bit moves 4.1-4.4 cycles, plain 32-bit arithmetic 2.3-2.7,
# transcendental 8.2
CLASS_CYCLES = {'packed': 4.2, 'plain': 2.5, 'trans': 8.2}


def cpu_baseline(spec, frames_dev, gpu_events, sample, threads):
    """Oracle on the first `sample` frames: C MOG + SimpleBlobDetector on one
    thread and threaded (rows / frames), then the Python steps 3-4.  Also the
    parity check."""
    from oracle import detect_ref as D
    from oracle import em_ref
    host = frames_dev[:sample].cpu().numpy()
    bp = D.hawkeye_params(spec['height'])
    t0 = time.perf_counter()
    events = D.detect_clip(host, bp, threads=1)
    t1 = time.perf_counter()
    many = D.detect_clip(host, bp, threads=threads)
    t2 = time.perf_counter()
    meas = {f: [(float(k[0]), float(k[1]), float(k[2])) for k in ev]
            for f, ev in enumerate(events)}
    sc = em_ref.Scene(meas, spec['width'], spec['height'], spec['fps'],
                      frame_count=sample)
    sc.build_initial_arcs()
    sc.em_optimize()
    t3 = time.perf_counter()

    def same(a, b):
        return len(a) == len(b) and all(
            np.array_equal(np.asarray(x).view(np.uint32),
                           np.asarray(y).view(np.uint32))
            for x, y in zip(a, b))
    one, thr, em = t1 - t0, t2 - t1, t3 - t2
    base = {
        'value': round(sample / (one + em), 3), 'unit': 'frames/s',
        'cores': 1, 'kind': 'port',
        'sample': 'first {} frames of the same clip (the 200-frame learning '
                  'ramp of MOG and {} frames after it): C oracle MOG + '
                  'SimpleBlobDetector {:.2f} s on one thread, Python oracle '
                  'steps 3-4 {:.2f} s'.format(sample, max(0, sample - 200),
                                              one, em),
        'detect_only_fps': round(sample / one, 3),
        'threaded': {
            'cores': threads,
            'detect_only_fps': round(sample / thr, 3),
            'value': round(sample / (thr + em), 3),
            'events_equal_single_thread': bool(same(events, many)),
            'note': 'MOG over bands of rows and blob detection over frames '
                    '(how OpenCV spreads them with parallel_for_), {:.2f} s; '
                    'steps 3-4 stay the single-thread Python port'.format(thr),
        },
    }
    parity = {'events_bit_exact_vs_oracle': bool(same(gpu_events[:sample],
                                                      events)),
              'frames_compared': sample,
              'events_compared': int(sum(len(e) for e in events))}
    return base, parity


def from_host_rate(scanner, spec, frames_dev, nframes, device):
    """The same hot path fed from page-locked host memory (what a host
    decoder hands over): chunks cross PCIe on a copy stream while the previous
    chunk is processed.  Reported beside `value`, never as `value`."""
    from hawkeye_amd.frames import PinnedHostSource
    n = min(nframes, frames_dev.shape[0])
    host = PinnedHostSource.alloc(n, spec['height'], spec['width'])
    host.copy_(frames_dev[:n])
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        src = PinnedHostSource(host, spec['fps'], device=device, chunk=CHUNK)
        t0 = time.perf_counter()
        events = scanner.detect(src)
        dt = time.perf_counter() - t0
        src.close()
        best = dt if best is None else min(best, dt)
    ref = scanner.detect(frames_dev[:n])
    same = all(np.array_equal(a, b) for a, b in zip(events, ref))
    gb = n * spec['height'] * spec['width'] * 3 / 1e9
    return {'frames': n, 'detect_fps': round(n / best, 1),
            'h2d_GBps': round(gb / best, 2),
            'events_equal_device_resident': bool(same),
            'note': 'step 2 only, frames read from pinned host memory '
                    'through hawkeye_amd.frames.PinnedHostSource'}


def mog_alone_ms(scanner, frames, size):
    """Average duration of MOG launches of `size` frames (the pipeline's
    chunk length) over the clip with the GPU to themselves, HIP events on the
    launch stream."""
    det = scanner.detector
    size = max(1, min(CHUNK, int(round(size))))
    full = frames.shape[0] // size
    if full == 0:
        return None
    det.mog.reset()
    masks = None
    if not det.use_tiles:
        masks = torch.empty((size, frames.shape[1], frames.shape[2]),
                            dtype=torch.uint8, device=frames.device)

    def launch(s):
        if det.use_tiles:
            det.mog.apply_batch_tiles(frames[s:s + size], det._tiles[0])
        else:
            det.mog.apply_batch(frames[s:s + size], out=masks)
    launch(0)                                   # model and launch order set up
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(1, full):
        launch(k * size)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / max(1, full - 1) if full > 1 else None


def profile_json(name):
    """A committed summary under profiles/ (written by
    tools/mog_profile_summary.py from rocprofv3 PMC passes), or None."""
    try:
        with open(os.path.join(REPO, 'profiles', name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def mog_roofline(launch_ms, alone_ms, launch_frames, px, scene):
    """The roofline object of mog_batch2_k for launches of `launch_frames`
    frames of `px` pixels that took `launch_ms` each (measured by the caller
    with events on the launch stream).  `scene`: which profiled scene of
    profiles/mog_issue.json the instruction counts come from ('clip' = the
    1080p benchmark clip) or None when the workload was not profiled."""
    sec = launch_ms * 1e-3
    issue = profile_json('mog_issue.json')
    traffic = profile_json('mog_traffic.json')
    counts = None
    if issue is not None and scene is not None:
        counts = issue.get('scenes', {}).get(scene)
    peak = SIMDS * NOMINAL_CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9
    roof = {'bound': 'valu', 'kernel': 'mog_batch2_k',
            'achieved': None, 'peak': round(peak, 1),
            'unit': 'G wave-instructions/s (VALU)', 'frac': None,
            'traffic': None}
    # the counters are per profiled launch; the kernel's work is per frame
    scale = launch_frames / float((issue or {}).get('frames_per_launch', 128))
    if counts is not None:
        # quad-cycle issue slots: one per VALU instruction, two for the few
        # transcendental ones (SQ_ACTIVE_INST_VALU counts exactly that)
        slots = counts['SQ_ACTIVE_INST_VALU']['per_launch'] * scale
        roof['achieved'] = round(slots / sec / 1e9, 1)
        roof['frac'] = round(slots / sec / 1e9 / peak, 4)
        if alone_ms:
            roof['frac_alone'] = round(
                slots / (alone_ms * 1e-3) / 1e9 / peak, 4)
        roof['valu_instructions_per_launch'] = int(
            counts['SQ_INSTS_VALU']['per_launch'] * scale)
        roof['valu_issue_slots_per_launch'] = int(slots)
        roof['valu_per_wave_frame'] = \
            counts['SQ_INSTS_VALU']['per_wave_frame']
        roof['salu_per_wave_frame'] = \
            counts['SQ_INSTS_SALU']['per_wave_frame']
        roof['valu_busy_under_profiler'] = counts['valu_busy']
        # the same with every class at its own measured price instead of the
        # 4-cycle upper bound: f32 add / mul / fma (packed in this kernel) and
        # the byte converts at the packed price, transcendental ones at
        # theirs, whatever is left (integer, compare, select, move, max) at
        # the plain price -- a lower estimate of the issue cycles used
        def per_wf(name):
            return counts.get(name, {}).get('per_wave_frame', 0.0)
        arith = per_wf('SQ_INSTS_VALU_ADD_F32') + \
            per_wf('SQ_INSTS_VALU_MUL_F32') + \
            per_wf('SQ_INSTS_VALU_FMA_F32') + per_wf('SQ_INSTS_VALU_CVT')
        trans = per_wf('SQ_INSTS_VALU_TRANS_F32')
        rest = max(0.0, per_wf('SQ_INSTS_VALU') - arith - trans)
        cycles = CLASS_CYCLES['packed'] * arith + \
            CLASS_CYCLES['trans'] * trans + CLASS_CYCLES['plain'] * rest
        wave_frames = (issue or {}).get('wave_frames_per_launch', 0) * scale
        if wave_frames:
            roof['frac_class_priced'] = round(
                cycles * wave_frames / (SIMDS * NOMINAL_CLOCK_HZ * sec), 4)
            roof['class_prices_cycles'] = CLASS_CYCLES
    if traffic is not None and scene == 'clip':
        hbm = traffic['bytes_per_launch'] * scale
        roof['traffic'] = int(hbm)
        roof['traffic_source'] = traffic.get('source')
        roof['hbm'] = {'achieved': round(hbm / sec / 1e9, 1),
                       'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                       'frac': round(hbm / sec / 1e9 / HBM_PEAK_GBS, 4),
                       'over_fetch_of_frame_lines': 1.65,
                       'note': 'bytes the L2 asked the fabric for (PMC) / '
                               'launch time: the second roof.  Counter scale '
                               'calibrated on this kernel\'s access shapes '
                               '(profiles/r03_hbm_counter_calibration.csv): '
                               'FETCH_SIZE counts a 128-byte request as 64 '
                               'bytes for the frame loads and the plane '
                               'loads alike (so it is doubled), WRITE_SIZE '
                               'is exact; the four 96-byte row segments of '
                               'a tile straddle 128-byte lines and the '
                               'frame lines are requested 1.65 times over '
                               '(Infinity-Cache hits included)'}
    algo = ALGO_BYTES_PER_PIXEL_FRAME * px * launch_frames
    roof['algorithmic'] = {
        'bytes_per_launch': int(algo),
        'GBps': round(algo / sec / 1e9, 1),
        'note': 'SURVEY.md 8d stream model (324 B per pixel per frame: the '
                'mixture read and written every frame).  The kernel keeps '
                'the mixture in registers and LDS over the frames of a '
                'launch, so this rate is above HBM peak and is not a '
                'roofline; see traffic for the bytes actually moved'}
    roof.update({
        'launch_ms': round(launch_ms, 4),
        'launch_ms_alone': None if alone_ms is None else round(alone_ms, 4),
        'frames_per_launch': int(launch_frames),
        'kernel_fps': round(launch_frames / sec, 1),
        'note': 'frac = VALU issue slots used / available: (VALU '
                'instructions of a launch, transcendental ones twice) x 4 '
                'cycles / (1024 SIMDs x 2.4 GHz x launch time).  Instruction '
                'counts: SQ counters of this kernel on this clip '
                '(profiles/mog_issue.json, tools/prof_mog_sq.sh); the 4 '
                'cycles per instruction per SIMD: counters of pure '
                'instruction streams (profiles/r03_ubench_counters.csv).  '
                'launch_ms = HIP events around the launches of the timed '
                'region, on their stream: the blob kernels of the previous '
                'chunk run beside them on another stream and share the '
                'chip, as they do under rocprofv3\'s kernel trace of the '
                'same command (profiles/r03_kernel_stats.csv); '
                'launch_ms_alone / frac_alone = the same launches with the '
                'GPU to themselves (what the counter passes profile)'})
    return roof


def launch_times(timing):
    """(average ms, frames) of the full-chunk MOG launches of a timed region
    from EventDetector.mog_timing."""
    ms = np.array([e0.elapsed_time(e1) for e0, e1, _ in timing])
    fr = np.array([n for _, _, n in timing])
    if len(ms) == 0:
        return None, None
    full = fr == CHUNK
    if full.any():
        return float(ms[full].mean()), CHUNK
    return float(ms.mean()), float(fr.mean())


def run_clips(args, world, rank, local):
    xdev = 'cuda' if args.backend == 'nccl' else 'cpu'  # exchange buffers
    spec = dict(synth.CLIPS[WORKLOAD])
    spec['seed'] += rank                    # one clip per GPU
    if args.frames:
        spec['frames'] = args.frames
    renderer = synth.ClipRenderer(spec, device=local)
    frames = renderer.render(0, spec['frames'])
    torch.cuda.synchronize()
    scanner = ClipScanner(spec['width'], spec['height'], spec['fps'],
                          device=local, chunk=CHUNK)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up, and the latency of one clip scanned on its own (no overlap
    # between the detection and the arc stages)
    for _ in range(args.warmup):
        result = scanner.scan(frames)
    # (the HIP runtime still grows its launch resources during the first two
    # or three clips -- single calls of 5 ms: the best of three is reported)
    single_ms = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        result = scanner.scan(frames)
        ms = (time.perf_counter() - t0) * 1e3
        if single_ms is None or ms < single_ms:
            single_ms, single_stages = ms, result.stage_seconds

    # K steps = K passes of the whole hot path over the clip, pipelined the
    # way a batch of clips is (ClipScanner.scan_many)
    builders, host_threads = ClipScanner.plan_host_stage(count=args.steps)
    if args.builders is not None:
        builders = args.builders
    if args.host_threads is not None:
        host_threads = args.host_threads
    # Untimed pre-roll of the pipelined path itself (the W warm-up scans above
    # run one clip at a time and leave the builder threads' streams, the
    # pooled EM workspaces and the chip's clocks cold): batches of W clips
    # until `--preroll-seconds` have passed.  A fresh box gives 5-10 % less in
    # its first second of pipelined work than afterwards (tools/probe/warm.py).
    preroll_t0 = time.perf_counter()
    preroll_clips = 0
    while args.warmup > 0 and \
            time.perf_counter() - preroll_t0 < args.preroll_seconds:
        n = max(2, args.warmup)
        for _ in scanner.scan_many((frames for _ in range(n)),
                                   host_threads=host_threads,
                                   builders=builders, count=n):
            preroll_clips += 1
    preroll_s = time.perf_counter() - preroll_t0
    # (every 8th launch is bracketed by timing events: bracketing all of them
    # slowed the timed region by 15 %, tools/probe/batch_stages.py)
    scanner.detector.mog_timing = []
    scanner.detector.mog_timing_every = args.time_every
    stage_s = {'detect': 0.0, 'step3': 0.0, 'step4': 0.0, 'step3_cpu': 0.0,
               'step4_cpu': 0.0}
    barrier()
    t0 = time.perf_counter()
    for result in scanner.scan_many((frames for _ in range(args.steps)),
                                    host_threads=host_threads,
                                    builders=builders, count=args.steps):
        for k in stage_s:
            stage_s[k] += result.stage_seconds[k]
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms, launch_frames = launch_times(scanner.detector.mog_timing)
    launches_timed = len(scanner.detector.mog_timing)
    scanner.detector.mog_timing = None
    scanner.detector.mog_timing_every = 1

    # the same launches with nothing else on the GPU (no blob kernels of the
    # previous chunk in their way): the kernel's own duration, which is what
    # rocprofv3's kernel trace reports
    alone_ms = mog_alone_ms(scanner, frames, launch_frames) \
        if rank == 0 and launch_frames else None

    per_frame, arc_params = result.events, result.arc_params
    tags_left, tilt = int(result.alive.sum()), result.camera_tilt
    # collect every rank's arc table on rank 0 (RCCL), and the slowest time
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n_arcs = torch.tensor([len(arc_params)], dtype=torch.int64,
                              device=xdev)
        counts = [torch.zeros_like(n_arcs) for _ in range(world)]
        dist.all_gather(counts, n_arcs)
        cap = int(max(c.item() for c in counts))
        mine = torch.zeros((cap, 5), dtype=torch.float64, device=xdev)
        mine[:len(arc_params)] = torch.from_numpy(arc_params).to(xdev)
        tables = [torch.zeros_like(mine) for _ in range(world)] \
            if rank == 0 else None
        dist.gather(mine, tables, dst=0)
        total_arcs = int(sum(c.item() for c in counts))
    else:
        total_arcs = len(arc_params)
    if rank != 0:
        return None

    total_frames = spec['frames'] * args.steps * world
    value = total_frames / elapsed
    px = spec['width'] * spec['height']
    out = {
        'metric': '1080p frames/sec (detect+EM)',
        'value': round(value, 1), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (MOG) / int64+f64 (blob) / f64 (EM)',
        'data': 'synthetic',
        'config': {'workload': WORKLOAD, 'layout': 'clips',
                   'width': spec['width'],
                   'height': spec['height'], 'fps': spec['fps'],
                   'frames_per_clip': spec['frames'],
                   'clips': world, 'sharding': 'one clip per GPU, arc tables '
                   'gathered to rank 0 over RCCL' if world > 1 else 'single GPU',
                   'detect_chunk_frames': CHUNK},
        'host_cores_per_rank': host_core_share(),
        'builders': builders, 'step3_threads_per_clip': host_threads,
        'detect_lanes': int(os.environ.get('HK_DETECT_LANES', '1')),
        'untimed_preroll': {'seconds': round(preroll_s, 2),
                            'pipelined_clips': preroll_clips,
                            'single_clip_scans': args.warmup + 3},
        'pipeline': 'detection of the passes ahead (GPU) beside steps 3-4 '
                    'of the passes before (one host thread per pass); every '
                    'pass does all the work',
        'stage_ms_per_step': {k: round(v / args.steps * 1e3, 3)
                              for k, v in stage_s.items()},
        'single_clip_ms': None if single_ms is None else round(single_ms, 3),
        'single_clip_fps': None if single_ms is None else round(
            spec['frames'] / (single_ms * 1e-3), 1),
        'single_clip_stage_ms': None if single_ms is None else {
            k: round(v * 1e3, 3) for k, v in single_stages.items()},
        'result': {'events': int(sum(len(k) for k in per_frame)),
                   'arcs': total_arcs, 'tags_on_arcs': tags_left,
                   'camera_tilt_rad': tilt},
        'roofline': mog_roofline(launch_ms, alone_ms, launch_frames, px,
                                 None if args.frames else 'clip'),
    }
    out['roofline']['launches_timed'] = launches_timed

    if world == 1 and args.host_frames > 0:
        out['pcie_inclusive'] = from_host_rate(scanner, spec, frames,
                                               args.host_frames, local)

    if world == 1 and not args.no_cpu_baseline:
        sample = min(args.cpu_sample, spec['frames'])
        base, parity = cpu_baseline(spec, frames, per_frame, sample,
                                    host_core_share())
        out['cpu_baseline'] = base
        out['parity'] = parity
        out['vs_cpu_baseline'] = {
            'single_clip_vs_one_core': round(
                out['single_clip_fps'] / base['value'], 1),
            'single_clip_vs_threaded': round(
                out['single_clip_fps'] / base['threaded']['value'], 1),
            'note': 'one clip on its own (GPU + this rank\'s host cores, no '
                    'overlap between clips) against the oracle on the '
                    'sample; `value` is batch throughput and is not divided '
                    'by a baseline here.  Steps 3-4 of the baseline are a '
                    'Python port, so these ratios flatter the arc stages'}
    return out


class ResidentSegment:
    """Frames [base, base + n) of a clip, already in HBM, as a frame source
    with absolute seek() (what hawkeye_amd.dist.detect_segment drives)."""

    def __init__(self, frames, base, spec):
        self._frames, self._base, self._pos = frames, base, 0
        self.fps = float(spec['fps'])
        self.width, self.height = spec['width'], spec['height']
        self.frame_count_estimate = spec['frames']

    def read(self, max_frames):
        out = self._frames[self._pos:self._pos + max_frames]
        self._pos += out.shape[0]
        return out

    def seek(self, frame):
        self._pos = int(frame) - self._base

    def close(self):
        pass


def nearest_match_rate(kept, ref, tol=0.5):
    """Share of the events of `ref` (per-frame arrays) that have an event of
    `kept` within `tol` pixels in the same frame."""
    matched = total = 0
    for a, b in zip(kept, ref):
        total += len(b)
        if len(a) and len(b):
            d = np.hypot(b[:, None, 0] - a[None, :, 0],
                         b[:, None, 1] - a[None, :, 1])
            matched += int((d.min(axis=1) <= tol).sum())
    return matched, total


def run_segments(args, world, rank, local):
    xdev = 'cuda' if args.backend == 'nccl' else 'cpu'  # exchange buffers
    from hawkeye_amd.frames import SyntheticSource
    spec = dict(synth.CLIPS[SEGMENT_WORKLOAD])
    if args.frames:
        spec['frames'] = args.frames
    fcount, warmup = spec['frames'], args.segment_warmup
    first, keep, end = hkdist.segment_bounds(fcount, world, warmup)[rank]
    frame_bytes = spec['width'] * spec['height'] * 3
    total_mem = torch.cuda.get_device_properties(local).total_memory
    # (a rehearsal with --share-gpu puts every rank's segment on the one GPU)
    sharers = world if args.share_gpu else 1
    resident = (end - first) * frame_bytes <= 0.72 * total_mem / sharers
    # all ranks take the same route, so that the line describes one thing
    if world > 1:
        flag = torch.tensor([1 if resident else 0], device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        resident = bool(flag.item())
    if resident:
        segment = synth.ClipRenderer(spec, device=local).render(
            first, end - first)
        torch.cuda.synchronize()

        def make_source():
            return ResidentSegment(segment, first, spec)
    else:
        def make_source():
            return SyntheticSource(spec, device=local)

    scanner = ClipScanner(spec['width'], spec['height'], spec['fps'],
                          device=local, chunk=CHUNK)
    det = scanner.detector

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass():
        t0 = time.perf_counter()
        events, _ = hkdist.detect_time_segmented(
            make_source, warmup=warmup, device=local, chunk=CHUNK,
            detector=det)
        t1 = time.perf_counter()
        res = None
        if rank == 0:
            res = scanner.refine_arcs(*scanner.build_arcs(events))
            res.stage_seconds['detect+gather'] = t1 - t0
        return res

    for _ in range(args.warmup):
        one_pass()
    det.mog_timing = []
    det.mog_timing_every = args.time_every
    stage_s = {'detect+gather': 0.0, 'step3': 0.0, 'step4': 0.0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_pass()
        if res is not None:
            for k in stage_s:
                stage_s[k] += res.stage_seconds[k]
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms, launch_frames = launch_times(det.mog_timing)
    det.mog_timing = None
    det.mog_timing_every = 1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # how close a cold start `warmup` frames early comes to the sequential
    # run: the last rank repeats the head of its segment with a four times
    # longer warm-up (the sequential run itself would cost as much as N = 1)
    agreement = None
    if world > 1 and rank == world - 1:
        n = min(256, end - keep)
        long_first = max(0, keep - 4 * warmup)
        src = SyntheticSource(spec, device=local)
        det.mog.reset()
        ref = hkdist.detect_segment(src, det, long_first, keep, keep + n,
                                    chunk=CHUNK)
        det.mog.reset()
        mine = hkdist.detect_segment(make_source(), det, first, keep,
                                     keep + n, chunk=CHUNK)
        m, tot = nearest_match_rate(mine, ref)
        agreement = [m, tot, n, keep - long_first]
    if world > 1:
        box = [agreement]
        dist.broadcast_object_list(box, src=world - 1)
        agreement = box[0]
    if rank != 0:
        return None

    px = spec['width'] * spec['height']
    bounds = hkdist.segment_bounds(fcount, world, warmup)
    extra = sum(k - f for f, k, _ in bounds)
    out = {
        'metric': '1080p frames/sec (detect+EM)',
        'value': round(fcount * args.steps / elapsed, 1), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32 (MOG) / int64+f64 (blob) / f64 (EM)',
        'data': 'synthetic',
        'config': {'workload': SEGMENT_WORKLOAD, 'layout': 'segments',
                   'width': spec['width'], 'height': spec['height'],
                   'fps': spec['fps'], 'frames_per_clip': fcount, 'clips': 1,
                   'sharding': 'contiguous time segments, one per GPU, each '
                               'started {} frames early with a cold MOG '
                               'model; kept events gathered on rank 0 over '
                               'RCCL; steps 3-4 on rank 0'.format(warmup),
                   'pixels_per_frame_vs_1080p': round(px / (1920 * 1080.), 2),
                   'detect_chunk_frames': CHUNK},
        'inputs': 'each rank\'s frames resident in HBM before the timed '
                  'region' if resident else
                  'frames rendered on the GPU chunk by chunk INSIDE the timed '
                  'region (the clip is {:.0f} GB and does not fit one GPU)'
                  .format(fcount * frame_bytes / 1e9),
        'host_cores_per_rank': host_core_share(),
        'builders': 1, 'step3_threads_per_clip': host_core_share(),
        'warmup_frames_per_segment': warmup,
        'warmup_overhead': round(extra / float(fcount), 4),
        'stage_ms_per_step': {k: round(v / args.steps * 1e3, 3)
                              for k, v in stage_s.items()},
        'result': {'events': int(sum(len(k) for k in res.events)),
                   'arcs': len(res.arc_params),
                   'tags_on_arcs': int(res.alive.sum()),
                   'camera_tilt_rad': res.camera_tilt},
        'event_agreement': None if agreement is None else {
            'matched_within_half_pixel': agreement[0],
            'events': agreement[1], 'frames': agreement[2],
            'rate': round(agreement[0] / max(1, agreement[1]), 5),
            'against': 'the same frames (head of the last segment) detected '
                       'after a {}-frame warm-up; against the sequential run '
                       'itself: tests/test_gpu_configs.py'.format(agreement[3])},
        'roofline': mog_roofline(launch_ms, None, launch_frames, px, None)
        if launch_ms else None,
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--layout', choices=('clips', 'segments'),
                    default='clips')
    ap.add_argument('--frames', type=int, default=None,
                    help='shorter clip (debugging only; invalidates value)')
    ap.add_argument('--segment-warmup', type=int, default=400,
                    help='frames a segment starts early (--layout segments)')
    ap.add_argument('--preroll-seconds', type=float, default=2.0,
                    help='untimed pipelined batches before the timed region '
                         '(--layout clips; 0 = none)')
    ap.add_argument('--time-every', type=int, default=8,
                    help='bracket every n-th MOG launch of the timed region '
                         'with timing events (roofline.launch_ms)')
    ap.add_argument('--backend', default='nccl',
                    help="torch.distributed backend for N > 1 ('nccl' = RCCL; "
                         "'gloo' only to rehearse the control flow)")
    ap.add_argument('--share-gpu', action='store_true',
                    help='every rank uses GPU 0 (rehearsal on a one-GPU box; '
                         'the numbers mean nothing)')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='start the ranks, form the process group, report its '
                         'size and stop before any GPU work (tests)')
    ap.add_argument('--cpu-sample', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-threads', type=int, default=None,
                    help='step-3 worker threads per clip in the pipelined '
                         'run (default: ClipScanner.plan_host_stage)')
    ap.add_argument('--builders', type=int, default=None,
                    help='clips in step 3 at the same time (pipelined run; '
                         'default: ClipScanner.plan_host_stage)')
    ap.add_argument('--host-frames', type=int, default=512,
                    help='frames of the PCIe-inclusive side measurement '
                         '(0 = skip)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 60 if args.layout == 'clips' else 3

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # not under a launcher: be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world
    load_runtime()
    if args.share_gpu:      # rehearsal of the N > 1 control flow on one GPU
        local = 0
    if args.rendezvous_only:
        dist.init_process_group(args.backend)
        seen = torch.tensor([1])
        dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({'rendezvous': int(seen.item()),
                              'world_size': dist.get_world_size(),
                              'backend': args.backend}))
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl',
                                    device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.backend)

    run = run_clips if args.layout == 'clips' else run_segments
    out = run(args, world, rank, local)
    if out is not None:
        # how many ranks the collective backend itself saw (RCCL for 'nccl')
        out['ranks_in_process_group'] = dist.get_world_size() \
            if dist.is_initialized() else 1
        out['backend'] = args.backend if world > 1 else None
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
