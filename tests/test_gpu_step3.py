"""
GPU: step 3 with its seeds evaluated on the device (hk_seed.hip, through the
C ABI) against the host code path, bit for bit: same arcs, same members, same
parameters; with the library's self check on, every single seed's device
result is compared with the host's.

The product library is built without the device side (it is the slower path);
`make DEVICE_SEEDS=1` builds hawkeye_amd/libhawkeye_hip_device_seeds.so with
it.  With the product library loaded the cases below skip, and
test_device_side_in_its_own_library runs this file once more in a child
process that loads the other library (HK_NATIVE_LIB).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def tags_with_distractors(width, height, fps, balls, frames, seed, clutter,
                          walkers):
    from hawkeye_amd import synth
    scene = synth.JugglingScene(width, height, fps, balls, frames, seed)
    meas = synth.make_tags(scene, seed + 1, clutter_per_frame=clutter)
    rng = np.random.RandomState(seed + 2)
    for d in range(walkers):                # random-walk distractors
        px, py = rng.uniform(0.1, 0.9) * width, rng.uniform(0.1, 0.9) * height
        for f in range(frames):
            px += rng.normal(0, 1.2)
            py += rng.normal(0, 1.2)
            meas.setdefault(f, []).append((px, py, 14.0 + d))
    return synth.tags_to_soa(meas, frames)


def build(monkeypatch, geometry, soa, device, window=None, self_check=False):
    from hawkeye_amd import _native as nat
    from hawkeye_amd.arcs import Clip
    if device and not nat.lib.hk_step3_device_built():
        pytest.skip('library built without the opt-in device side of step 3 '
                    '(make -C hawkeye_amd/csrc DEVICE_SEEDS=1, then '
                    'HK_NATIVE_LIB=hawkeye_amd/libhawkeye_hip_device_seeds.so)')
    width, height, fps, frames = geometry
    frame, x, y, r, tw, start = soa
    monkeypatch.setenv('HK_STEP3_DEVICE', '2' if device else '0')
    if window:
        monkeypatch.setenv('HK_STEP3_DEVICE_WINDOW', str(window))
    else:
        monkeypatch.delenv('HK_STEP3_DEVICE_WINDOW', raising=False)
    nat.check(nat.lib.hk_set_self_check(1 if self_check else 0))
    try:
        clip = Clip(width, height, fps, frames, frame, x, y, r, tw, stream=0)
        clip.build_initial_arcs()
        params, ids, tags = clip.arcs()
        g = clip.state()['g_px_per_frame_sq']
        clip.close()
    finally:
        nat.check(nat.lib.hk_set_self_check(0))
    return (params.tobytes(), ids.tobytes(), tuple(t.tobytes() for t in tags),
            g)


@pytest.mark.parametrize('geometry,balls,clutter,walkers,seed', [
    ((1280, 720, 60.0, 400), 7, 15.0, 6, 901),
    ((1920, 1080, 60.0, 600), 9, 6.0, 20, 17),
    ((960, 540, 30.0, 300), 3, 2.0, 0, 5),
])
def test_device_seeds_equal_host_seeds(monkeypatch, geometry, balls, clutter,
                                       walkers, seed):
    soa = tags_with_distractors(geometry[0], geometry[1], geometry[2], balls,
                                geometry[3], seed, clutter, walkers)
    want = build(monkeypatch, geometry, soa, device=False)
    assert len(want[2]) > 10
    # every seed cross-checked inside the library, several window sizes (the
    # smaller the window, the fewer results go stale)
    assert build(monkeypatch, geometry, soa, True, self_check=True) == want
    for window in (64, 1000, 100000):
        assert build(monkeypatch, geometry, soa, True, window=window) == want


def test_device_path_on_golden_run(monkeypatch, golden_dir):
    """The 1080p60 golden run of the reference through the device path."""
    import json
    import os
    from hawkeye_amd import synth
    with open(os.path.join(golden_dir, 'run_noisy_1080p60_9ball.json')) as fh:
        run = json.load(fh)
    meas = {int(f): [tuple(t) for t in lst] for f, lst in run['meas'].items()}
    soa = synth.tags_to_soa(meas, run['frames'])
    geometry = (run['width'], run['height'], run['fps'], run['frames'])
    host = build(monkeypatch, geometry, soa, device=False)
    dev = build(monkeypatch, geometry, soa, device=True, self_check=True)
    assert dev == host
    assert len(dev[2]) == len(run['step3']['arcs'])


def test_device_side_in_its_own_library():
    from hawkeye_amd import _native as nat
    if nat.lib.hk_step3_device_built():
        pytest.skip('this process has the device side loaded: the cases above ran')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    other = os.path.join(repo, 'hawkeye_amd', 'libhawkeye_hip_device_seeds.so')
    if not os.path.exists(other):
        pytest.skip('hawkeye_amd/libhawkeye_hip_device_seeds.so is not built '
                    '(make -C hawkeye_amd/csrc DEVICE_SEEDS=1)')
    env = dict(os.environ, HK_NATIVE_LIB=other)
    out = subprocess.run(
        [sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x',
         '-m', 'gpu', '-p', 'no:cacheprovider',
         '--deselect', 'tests/test_gpu_step3.py::test_device_side_in_its_own_library'],
        cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    tail = out.stdout[-2000:] + out.stderr[-2000:]
    assert out.returncode == 0, tail
    assert ' passed' in out.stdout and 'skipped' not in out.stdout.splitlines()[-1], tail
