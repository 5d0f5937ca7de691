"""
CPU: native step 3 (host C++ behind the C ABI, no GPU involved) against the
golden runs produced by the reference, and against the Python oracle on a
fresh seeded input.
"""
import json
import os

import numpy as np
import pytest

from hawkeye_amd import synth
from hawkeye_amd.arcs import Clip

RUNS = ['clean_720p30_5ball', 'noisy_1080p60_9ball', 'tilt3_720p30_5ball',
        'lowfps_480p24_3ball']
REL = 1e-6      # see tests/test_oracle_em.py for why not tighter


def close(a, b, rel=REL, abs_=1e-9):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


def load_run(golden_dir, name):
    with open(os.path.join(golden_dir, 'run_{}.json'.format(name))) as fh:
        run = json.load(fh)
    meas = {int(f): [tuple(t) for t in lst] for f, lst in run['meas'].items()}
    return run, meas


def make_clip(run, meas, **kw):
    frame, x, y, r, tw, start = synth.tags_to_soa(meas, run['frames'])
    clip = Clip(run['width'], run['height'], run['fps'], run['frames'], frame,
                x, y, r, tw, **kw)
    # (frame, index in frame) key of every tag, as the fixtures use
    key = [(int(f), int(i - start[f])) for i, f in enumerate(frame)]
    return clip, key


@pytest.mark.parametrize('name', RUNS)
def test_build_initial_arcs_matches_reference(golden_dir, name):
    run, meas = load_run(golden_dir, name)
    clip, key = make_clip(run, meas, stream=0)
    clip.build_initial_arcs()
    st = clip.state()
    want = run['step3']
    assert close(st['g_px_per_frame_sq'], want['g_px_per_frame_sq'])
    assert close(st['cm_per_pixel'], want['cm_per_pixel'])
    params, ids, tags = clip.arcs()
    assert len(params) == len(want['arcs'])
    for p, i, t, w in zip(params, ids, tags, want['arcs']):
        assert i == w['id']
        assert [list(key[j]) for j in t] == w['tags']
        for got, k in zip(p, ('a', 'b', 'c', 'e', 'f_peak')):
            assert close(got, w[k]), (k, i)


def test_matches_oracle_on_new_input():
    from oracle import em_ref
    scene = synth.JugglingScene(960, 540, 50.0, 4, 150, 77)
    meas = synth.make_tags(scene, 78, clutter_per_frame=5.0)
    frame, x, y, r, tw, start = synth.tags_to_soa(meas, 150)
    clip = Clip(960, 540, 50.0, 150, frame, x, y, r, tw, stream=0)
    clip.build_initial_arcs()
    sc = em_ref.Scene(meas, 960, 540, 50.0, frame_count=150)
    sc.build_initial_arcs()
    params, ids, tags = clip.arcs()
    assert len(params) == len(sc.arcs) > 5
    assert close(clip.state()['g_px_per_frame_sq'], sc.g)
    for p, t, arc in zip(params, tags, sc.arcs):
        assert list(t) == arc.tags
        for got, want in zip(p, arc.params()):
            assert close(got, want)


def test_progress_callback_can_abort(golden_dir):
    run, meas = load_run(golden_dir, 'noisy_1080p60_9ball')
    clip, _ = make_clip(run, meas, stream=0)
    calls = []

    class Stop(Exception):
        pass

    def cb():
        calls.append(1)
        if len(calls) == 3:
            raise Stop('enough')
    clip.set_progress(cb)
    with pytest.raises(Stop):
        clip.build_initial_arcs()
    assert len(calls) == 3


def test_rejects_unsorted_frames():
    from hawkeye_amd import _native as nat
    with pytest.raises(nat.NativeError, match='ascending'):
        Clip(64, 48, 30.0, 10, [3, 2], [1.0, 2.0], [1.0, 2.0], [5.0, 5.0],
             stream=0)


def test_result_does_not_depend_on_thread_count_or_window(monkeypatch):
    """The speculative builder must give the sequential result bit for bit:
    same arcs for 1 thread (no speculation at all), several threads, and a
    speculation window from minimal to very deep.  Input with clutter and
    slowly moving distractor tracks, so that many seeds grow long trial arcs
    and accepted arcs invalidate speculative results."""
    from hawkeye_amd import _native as nat
    scene = synth.JugglingScene(1280, 720, 60.0, 7, 400, 901)
    meas = synth.make_tags(scene, 902, clutter_per_frame=15.0)
    rng = np.random.RandomState(903)
    for d in range(6):                      # random-walk distractors
        px, py = rng.uniform(100, 1100), rng.uniform(100, 600)
        for f in range(400):
            px += rng.normal(0, 1.2)
            py += rng.normal(0, 1.2)
            meas.setdefault(f, []).append((px, py, 14.0 + d))
    frame, x, y, r, tw, start = synth.tags_to_soa(meas, 400)

    def run(threads, window):
        monkeypatch.setenv('HK_STEP3_WINDOW', str(window))
        nat.check(nat.lib.hk_set_host_threads(threads))
        try:
            clip = Clip(1280, 720, 60.0, 400, frame, x, y, r, tw, stream=0)
            clip.build_initial_arcs()
            params, ids, tags = clip.arcs()
            g = clip.state()['g_px_per_frame_sq']
            clip.close()
        finally:
            nat.check(nat.lib.hk_set_host_threads(0))
        return (params.tobytes(), ids.tobytes(),
                tuple(t.tobytes() for t in tags), g)

    want = run(1, 16)
    assert len(want[2]) > 20
    for threads, window in ((2, 1), (4, 16), (8, 256), (16, 1024), (3, 7)):
        assert run(threads, window) == want, (threads, window)
    # the AVX2 member / frontier loops against the scalar ones
    monkeypatch.setenv('HK_STEP3_SIMD', '0')
    assert run(1, 16) == want
    assert run(4, 64) == want


def test_helper_threads_do_not_change_the_result():
    """Threads without a clip of their own help running builds
    (hk_step3_help, what scan_many's idle builder threads do): three clips
    built at the same time on one thread each, four helpers hopping between
    them with small seed budgets; every result equals the unhelped one."""
    import threading
    from hawkeye_amd import _native as nat
    inputs = []
    for k in range(3):
        scene = synth.JugglingScene(1280, 720, 60.0, 5 + k, 300, 950 + k)
        meas = synth.make_tags(scene, 960 + k, clutter_per_frame=12.0)
        rng = np.random.RandomState(970 + k)
        for d in range(5):
            px, py = rng.uniform(100, 1100), rng.uniform(100, 600)
            for f in range(300):
                px += rng.normal(0, 1.2)
                py += rng.normal(0, 1.2)
                meas.setdefault(f, []).append((px, py, 14.0 + d))
        inputs.append(synth.tags_to_soa(meas, 300))

    def build(soa):
        frame, x, y, r, tw, start = soa
        clip = Clip(1280, 720, 60.0, 300, frame, x, y, r, tw, stream=0)
        clip.build_initial_arcs()
        params, ids, tags = clip.arcs()
        out = (params.tobytes(), ids.tobytes(),
               tuple(t.tobytes() for t in tags))
        clip.close()
        return out

    nat.check(nat.lib.hk_set_host_threads(1))
    try:
        want = [build(soa) for soa in inputs]
        assert all(len(w[2]) > 10 for w in want)
        for budget in (7, 2048):
            got = [None] * len(inputs)
            running = [len(inputs)]
            helped = [0]
            lock = threading.Lock()

            def builder(k):
                got[k] = build(inputs[k])
                with lock:
                    running[0] -= 1

            def helper():
                while True:
                    with lock:
                        if running[0] == 0:
                            return
                    if nat.lib.hk_step3_help(budget):
                        with lock:
                            helped[0] += 1

            threads = [threading.Thread(target=builder, args=(k,))
                       for k in range(len(inputs))] + \
                      [threading.Thread(target=helper) for _ in range(4)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert got == want, budget
            assert helped[0] > 0
    finally:
        nat.check(nat.lib.hk_set_host_threads(0))
    assert nat.lib.hk_step3_help(100) == 0      # nothing to help with


def test_random_tag_sets_match_oracle():
    """A short run of tests/fuzz_arcs.py (step 3 only, host code): random
    geometries, frame rates, ball counts, clutter and distractor tracks; the
    oracle rounds its powers like the native code (see em_ref), which leaves
    only the logic to compare."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        'fuzz_arcs.py')
    spec = importlib.util.spec_from_file_location('fuzz_arcs', path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from oracle import em_ref
    try:
        rng = np.random.RandomState(2024)
        done = 0
        while done < 3:
            case, meas = fz.random_case(rng)
            if sum(len(v) for v in meas.values()) > 1500:
                continue            # keep the Python oracle to seconds
            err = fz.compare(case, meas, False)
            assert err is None, (err, case)
            done += 1
    finally:
        em_ref.POWERS_AS_PRODUCTS = False


def test_equal_heights_and_positions_keep_tag_order():
    """Seeds are taken top of the frame first, tags of equal height in tag
    order (reference tracker.py:486-488 sorts stably by y), and the per-frame
    x index keeps equal x in tag order.  The native code orders the tags with
    a radix sort on the bit patterns of y and insertion sorts on x: whole-pixel
    coordinates (many ties), a shift that makes some of them negative, and
    -0.0 beside 0.0 must give the oracle's arcs."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        'fuzz_arcs.py')
    spec = importlib.util.spec_from_file_location('fuzz_arcs', path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from oracle import em_ref
    try:
        rng = np.random.RandomState(77)
        done = 0
        while done < 3:
            case, meas = fz.random_case(rng)
            total = sum(len(v) for v in meas.values())
            if total > 1500 or total < 300:     # radix path, seconds of oracle
                continue
            shift = [0.0, 40.0, 300.0][done]
            quant = {}
            for f, tags in meas.items():
                quant[f] = []
                for t in tags:
                    y = float(np.round(t[1] / 2.0) * 2.0) - shift
                    if y == 0.0 and len(quant[f]) % 2:
                        y = -0.0
                    quant[f].append((float(np.round(t[0])), y) + tuple(t[2:]))
            err = fz.compare(case, quant, False)
            assert err is None, (err, case, shift)
            done += 1
    finally:
        em_ref.POWERS_AS_PRODUCTS = False
