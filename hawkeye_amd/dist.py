"""
Multi-GPU layouts for the hot path, one process per GPU with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in CPU tests).

* **Clips** are independent: shard them over ranks, no data-path collective
  (``shard_clips``); only results are collected.
* **One long clip** can be time-segmented: rank r runs step 2 on frames
  ``[start_r - warmup, end_r)`` with a cold MOG model and drops the events of
  the warm-up frames; the per-rank event lists are then gathered to rank 0
  (one all_gather of counts + one grouped send/recv of 16-byte records), where
  steps 3-4 run on the whole clip.  This is approximate by nature: MOG is a
  recursive filter with a 200-frame time constant and a learning rate that
  depends on the global frame index for the first 200 frames, so a segment's
  masks differ slightly from the sequential run's near its head.  Measured
  mask agreement with the sequential run: 0.9994-0.9996 of the
  pixels after a 200-frame warm-up, >= 0.9998 after 400 (the default here),
  0.99994 after 600; ``tests/test_gpu_dist.py`` checks it.  Blob detection is
  per-frame exact.

The exchange is tiny (<= ~50 events x 16 B per frame) and latency-bound, so a
direct send to rank 0 inside one group call is used, not a ring collective.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_clips(n_clips, rank, world):
    """Indices of the clips rank `rank` owns (round-robin)."""
    return list(range(rank, n_clips, world))


def segment_bounds(frame_count, world, warmup):
    """[(first_frame_to_process, first_frame_to_keep, end_frame)] per rank:
    contiguous, near-equal segments; every segment but the first starts
    `warmup` frames early."""
    out = []
    for r in range(world):
        start = (frame_count * r) // world
        end = (frame_count * (r + 1)) // world
        out.append((max(0, start - warmup), start, end))
    return out


def pack_events(per_frame, first_frame):
    """Per-frame float32 [n][3] keypoint arrays -> int32 [m][4] records
    (frame, bits(x), bits(y), bits(size))."""
    counts = np.fromiter((len(k) for k in per_frame), np.int64, len(per_frame))
    m = int(counts.sum())
    rec = np.empty((m, 4), np.int32)
    if m:
        rec[:, 0] = np.repeat(np.arange(first_frame,
                                        first_frame + len(per_frame),
                                        dtype=np.int32), counts)
        kp = np.concatenate([np.asarray(k, np.float32).reshape(-1, 3)
                             for k in per_frame])
        rec[:, 1:] = kp.view(np.int32)
    return rec


def unpack_events(rec, frame_count):
    """Inverse of pack_events over the whole clip: a list with one float32
    [n][3] array per frame (records must be sorted by frame)."""
    rec = np.asarray(rec, np.int32).reshape(-1, 4)
    frames = rec[:, 0]
    if len(frames) and (np.diff(frames) < 0).any():
        raise ValueError('event records are not in frame order')
    counts = np.bincount(frames, minlength=frame_count)
    edges = np.concatenate([[0], np.cumsum(counts)])
    kp = np.ascontiguousarray(rec[:, 1:]).view(np.float32)
    return [kp[edges[f]:edges[f + 1]].copy() for f in range(frame_count)]


def gather_events(records, device=None, group=None):
    """Gather every rank's int32 [m][4] records on rank 0, concatenated in
    rank order (= frame order for time segments).  Returns the array on rank 0
    and None elsewhere.  `device`: where the exchange buffers live ('cuda' for
    the nccl/RCCL backend, 'cpu' for gloo)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if device is None:
        device = torch.device('cuda', default_device()) \
            if dist.get_backend(group) == 'nccl' else 'cpu'
    mine = torch.from_numpy(np.ascontiguousarray(records, np.int32)).to(device)
    count = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]

    ops, parts = [], None
    if rank == 0:
        parts = [mine] + [torch.empty((counts[r], 4), dtype=torch.int32,
                                      device=device)
                          for r in range(1, world)]
        for r in range(1, world):
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, parts[r], r, group=group))
    elif counts[rank]:
        ops.append(dist.P2POp(dist.isend, mine, 0, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != 0:
        return None
    return torch.cat(parts).cpu().numpy()


def detect_segment(source, detector, first, keep_from, end, chunk=128):
    """Step-2 arithmetic on frames [first, end) of `source` (which must
    support seek), returning the per-frame keypoints of [keep_from, end)."""
    source.seek(first)
    out = []
    pos = first
    while pos < end:
        frames = source.read(min(chunk, end - pos))
        if frames.shape[0] == 0:
            break
        out.extend(detector.submit(frames))
        pos += frames.shape[0]
    out.extend(detector.flush())
    return out[keep_from - first:]


def default_device():
    """The GPU of this rank: LOCAL_RANK when the launcher exported it (one
    process per GPU), else torch's current device."""
    local = os.environ.get('LOCAL_RANK')
    if local is not None:
        return int(local)
    return torch.cuda.current_device()


def detect_time_segmented(make_source, warmup=400, device=None, group=None,
                          chunk=128, detector=None):
    """Step 2 of one clip over all ranks of `group`: rank r detects frames
    [start_r - warmup, end_r) of `make_source()` (a frame source with seek())
    on its own GPU and the kept events are gathered on rank 0.  Returns
    (per-frame keypoint list of the whole clip, (fps, width, height)) on rank
    0 and (None, meta) elsewhere.  `device`: this rank's GPU (default: LOCAL_RANK,
    else the current device); the RCCL exchange buffers live on it."""
    from .detect import EventDetector, hawkeye_blob_params
    alone = not dist.is_initialized()       # a single process: no exchange
    world = 1 if alone else dist.get_world_size(group)
    rank = 0 if alone else dist.get_rank(group)
    if device is None:
        device = default_device()
    src = make_source()
    fcount = src.frame_count_estimate
    first, keep, end = segment_bounds(fcount, world, warmup)[rank]
    det = detector if detector is not None else EventDetector(
        src.width, src.height, blob_params=hawkeye_blob_params(src.height),
        chunk=chunk, device=device)
    try:
        det.mog.reset()
        per_frame = detect_segment(src, det, first, keep, end, chunk=chunk)
    finally:
        if detector is None:
            det.close()
    meta = (src.fps, src.width, src.height)
    src.close()
    if alone:
        return per_frame, meta
    where = torch.device('cuda', device) \
        if dist.get_backend(group) == 'nccl' else 'cpu'
    rec = gather_events(pack_events(per_frame, keep), device=where,
                        group=group)
    if rank != 0:
        return None, meta
    return unpack_events(rec, fcount), meta


def scan_time_segmented(make_source, warmup=400, device=None, group=None,
                        callback=None):
    """Steps 1-4 of one clip over all ranks of `group`: segmented step 2,
    event gather, steps 3-4 on rank 0.  Returns the notes dict on rank 0 (the
    same keys VideoScanner.process(steps=(1, 4)) leaves) and None elsewhere.
    `make_source()` -> a frame source with seek().  Every rank must name its
    own GPU: `device`, else LOCAL_RANK, else torch's current device."""
    from .tracker import VideoScanner
    from .types import Balltag
    if device is None:
        device = default_device()
    events, meta = detect_time_segmented(make_source, warmup=warmup,
                                         device=device, group=group)
    if events is None:
        return None
    fcount = len(events)
    sc = VideoScanner('time-segmented clip', device=device)
    notes = sc.notes
    notes['fps'], notes['frame_width'], notes['frame_height'] = meta
    notes['frame_count_estimate'] = notes['frame_count'] = fcount
    sp = notes['scanner_params']
    sp['min_tags_per_arc'] = (sp['min_tags_per_arc_high_fps']
                              if meta[0] >= 29
                              else sp['min_tags_per_arc_low_fps'])
    sp['max_distance_pixels'] = sp['max_distance_pixels_480'] * meta[2] / 480
    sp['radius_window'] = (sp['radius_window_high_res'] if meta[2] >= 480
                           else sp['radius_window_low_res'])
    notes['camera_tilt'] = 0.0
    notes['meas'] = {
        f: [Balltag(f, float(k[0]), float(k[1]), float(k[2]), 1.0)
            for k in events[f]] for f in range(fcount)}
    notes['step'] = 2
    sc.process(steps=(3, 4), callback=callback)
    return notes
