"""
CPU, world_size 2, gloo: the exchange step of the time-segmented multi-GPU
layout (segment bounds, record packing, gather to rank 0) and clip sharding.
No GPU and no native library involved.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_dist_module():
    # hawkeye_amd/dist.py only needs numpy + torch at import time
    spec = importlib.util.spec_from_file_location(
        'hk_dist', os.path.join(REPO, 'hawkeye_amd', 'dist.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fake_events(frame_count, seed):
    rng = np.random.RandomState(seed)
    return [rng.uniform(0, 2000, size=(rng.randint(0, 7), 3)).astype(
        np.float32) for _ in range(frame_count)]


def _worker(rank, world, port, frame_count, warmup, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        hk = load_dist_module()
        full = fake_events(frame_count, 5)
        first, keep, end = hk.segment_bounds(frame_count, world, warmup)[rank]
        # this rank "detects" [first, end) and keeps [keep, end)
        mine = full[first:end][keep - first:]
        rec = hk.gather_events(hk.pack_events(mine, keep), device='cpu')
        if rank == 0:
            got = hk.unpack_events(rec, frame_count)
            ok = len(got) == frame_count and all(
                np.array_equal(a.view(np.uint32), b.view(np.uint32))
                for a, b in zip(got, full))
            with open(out_path, 'w') as fh:
                fh.write('ok' if ok else 'mismatch')
        else:
            assert rec is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('frame_count,warmup', [(101, 20), (64, 200), (7, 0)])
def test_event_gather_world2(tmp_path, frame_count, warmup):
    out = str(tmp_path / 'result.txt')
    port = 29500 + (os.getpid() + frame_count) % 2000
    mp.spawn(_worker, args=(2, port, frame_count, warmup, out), nprocs=2,
             join=True)
    assert open(out).read() == 'ok'


def test_segment_bounds_and_sharding():
    hk = load_dist_module()
    b = hk.segment_bounds(14400, 8, 200)
    assert b[0] == (0, 0, 1800) and b[1] == (1600, 1800, 3600)
    assert b[-1] == (12400, 12600, 14400)
    assert [e for _, _, e in b][:-1] == [s for _, s, _ in b][1:]
    assert hk.segment_bounds(10, 3, 50) == [(0, 0, 3), (0, 3, 6), (0, 6, 10)]
    assert hk.shard_clips(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((hk.shard_clips(10, r, 4) for r in range(4)), [])) == \
        list(range(10))


def test_pack_roundtrip_and_order_check():
    hk = load_dist_module()
    ev = fake_events(30, 1)
    rec = hk.pack_events(ev[10:], 10)
    assert rec.shape[1] == 4 and rec[:, 0].min() >= 10
    back = hk.unpack_events(rec, 30)
    assert all(len(b) == 0 for b in back[:10])
    assert all(np.array_equal(a, b) for a, b in zip(back[10:], ev[10:]))
    with pytest.raises(ValueError):
        hk.unpack_events(rec[::-1], 30)
    assert hk.unpack_events(np.zeros((0, 4), np.int32), 3)[2].shape == (0, 3)


def test_host_core_share_and_stage_plan(monkeypatch):
    """A rank plans on its share of the node's cores; torch.distributed.run's
    default OMP_NUM_THREADS=1 is not taken as the machine's size."""
    from hawkeye_amd import batch
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(128)),
                        raising=False)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    monkeypatch.setenv('OMP_NUM_THREADS', '1')
    assert batch.host_core_share() == 16
    monkeypatch.setenv('OMP_NUM_THREADS', '12')
    assert batch.host_core_share() == 12
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '1')
    monkeypatch.setenv('OMP_NUM_THREADS', '1')
    assert batch.host_core_share() == 1
    monkeypatch.delenv('OMP_NUM_THREADS')
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(6)),
                        raising=False)
    assert batch.host_core_share() == 6
    plan = batch.ClipScanner.plan_host_stage
    assert plan(None, cores=16) == (15, 1)
    assert plan(4, cores=16) == (15, 1)
    for count in (None, 1, 20, 100):
        for cores in (1, 2, 3, 8, 16):
            b, t = plan(count, cores=cores)
            assert b >= 1 and t >= 1 and b * t <= max(1, cores - 1)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it: the parent
    starts two ranks (torch.distributed.run), they form the process group
    (gloo here: no GPU) and rank 0's line comes back on the parent's stdout
    (reference: none -- the reference has no distributed layer)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT')}
    out = subprocess.run(
        [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2',
         '--backend', 'gloo', '--share-gpu', '--rendezvous-only'],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    got = json.loads(lines[0])
    assert got == {'rendezvous': 2, 'world_size': 2, 'backend': 'gloo'}


def test_bench_parent_exits_with_the_ranks_status():
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run(
        [sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2',
         '--backend', 'no-such-backend', '--rendezvous-only'],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
