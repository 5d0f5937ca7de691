"""
GPU parity for step 4: E-step / M-step / tilt / distance kernels and the full
EM_optimize (through the C ABI) against golden vectors produced by the
reference, and against the Python oracle on fresh inputs.

Tolerance: fp64 everywhere; summation order differs (wave reduction vs the
reference's set order), and the determinants cancel ~7 digits, so arc
parameters are compared at 1e-6 relative (north_star: 1e-3); memberships and
hard assignments must be identical.
"""
import json
import os

import numpy as np
import pytest

from hawkeye_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

REL = 1e-6
RUNS = ['clean_720p30_5ball', 'noisy_1080p60_9ball', 'tilt3_720p30_5ball',
        'lowfps_480p24_3ball']


def close(a, b, rel=REL, abs_=1e-9):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


def meas_of(d):
    return {int(f): [tuple(t) for t in lst] for f, lst in d.items()}


def make_clip(meas, width, height, fps, frames):
    from hawkeye_amd.arcs import Clip
    frame, x, y, r, tw, start = synth.tags_to_soa(meas, frames)
    clip = Clip(width, height, fps, frames, frame, x, y, r, tw)
    key = [(int(f), int(i - start[f])) for i, f in enumerate(frame)]
    return clip, key


def test_em_operations_match_reference_unit_vectors(golden_dir):
    with open(os.path.join(golden_dir, 'unit_vectors.json')) as fh:
        u = json.load(fh)['estep']
    clip, key = make_clip(meas_of(u['meas']), u['width'], u['height'],
                          u['fps'], u['frames'])
    clip.build_initial_arcs()
    clip.estimate_camera_tilt()
    assert close(clip.state()['camera_tilt'], u['camera_tilt'])
    clip.fit_arcs()
    params, ids, tags = clip.arcs()
    for p, want in zip(params, u['arcs_after_fit']):
        for got, k in zip(p, ('a', 'b', 'c', 'e', 'f_peak')):
            assert close(got, want[k]), k

    # K7 against the host formula
    d = clip.member_distances()
    tilt = clip.state()['camera_tilt']
    s, c = np.sin(tilt), np.cos(tilt)
    pos = 0
    for p, t in zip(params, tags):
        tt = clip.frame[t] - p[4]
        xj, yj = p[0] + p[1] * tt, p[2] + p[3] * tt * tt
        px, py = xj * c + yj * s, yj * c - xj * s
        want = np.sqrt((px - clip.x[t])**2 + (py - clip.y[t])**2)
        assert np.allclose(d[pos:pos + len(t)], want, rtol=1e-13, atol=1e-12)
        pos += len(t)

    clip.calculate_weights()
    tag, arc, w = clip.weights()
    got = sorted([key[t][0], key[t][1], int(a), float(x)]
                 for t, a, x in zip(tag, arc, w))
    want = sorted(u['weights'])
    assert len(got) == len(want)
    for g, x in zip(got, want):
        assert g[:3] == x[:3] and close(g[3], x[3], rel=1e-12)
    _, _, tags = clip.arcs()
    for t, members in zip(tags, u['members']):
        assert [list(key[j]) for j in t] == members
    for k, codes in enumerate(u['eval_codes']):
        assert [clip.eval_arc(k, False), clip.eval_arc(k, True)] == codes


def test_m_step_skip_paths_match_the_reference(golden_dir):
    """K6 arc_moments_k through hk_clip_set_arcs + hk_clip_fit_arcs on the
    reference's own fit_arcs() results (tests/golden/make_golden_fit1.py, unit
    weights): the fits, and every way the reference leaves an arc untouched
    (reference tracker.py:714-715 fewer than 3 tags, :757-760 singular normal
    equations, :786-788 NaN or e == 0)."""
    from hawkeye_amd.arcs import Clip
    with open(os.path.join(golden_dir, 'fit_unit_vectors.json')) as fh:
        cases = json.load(fh)['fit_arcs_unit_weights']
    assert sum(c['skipped'] for c in cases) >= 5
    for case in cases:
        tags = case['tags']
        clip = Clip(1280, 720, 30.0, 200, [t[0] for t in tags],
                    [t[1] for t in tags], [t[2] for t in tags],
                    [16.0] * len(tags))
        clip.set_state(camera_tilt=case['tilt'])
        clip.set_arcs([case['start']], [list(range(len(tags)))])
        clip.fit_arcs()
        params, _, members = clip.arcs()
        assert len(members[0]) == len(tags)
        got = [float(v) for v in params[0]]
        if case['skipped']:
            assert got == case['start'] == case['result'], case['name']
        else:
            for g, w in zip(got, case['result']):
                assert close(g, w), (case['name'], g, w)
        clip.close()


@pytest.mark.parametrize('name', RUNS)
def test_em_optimize_matches_reference(golden_dir, name):
    with open(os.path.join(golden_dir, 'run_{}.json'.format(name))) as fh:
        run = json.load(fh)
    clip, key = make_clip(meas_of(run['meas']), run['width'], run['height'],
                          run['fps'], run['frames'])
    clip.build_initial_arcs()
    clip.em_optimize()
    want = run['step4']
    assert close(clip.state()['camera_tilt'], want['camera_tilt'])
    params, ids, tags = clip.arcs()
    assert len(params) == len(want['arcs'])
    for p, i, w in zip(params, ids, want['arcs']):
        assert i == w['id']
        for got, k in zip(p, ('a', 'b', 'c', 'e', 'f_peak')):
            assert close(got, w[k]), (k, i)
    alive, final = clip.tag_state()
    got = [[key[i][0], key[i][1], int(final[i])] for i in range(clip.n)
           if alive[i]]
    assert got == want['assign']
    # arc.tags after step 4 == hard assignment
    for k, t in enumerate(tags):
        assert all(final[j] == k for j in t)
    assert sum(len(t) for t in tags) == want['tags_remaining']


@pytest.mark.parametrize('spec', [
    (1920, 1080, 60.0, 9, 420, 501, 20.0, 0.0),     # BASELINE-like, noisy
    (1280, 720, 30.0, 7, 300, 502, 8.0, -2.0),
])
def test_em_optimize_matches_oracle_on_new_input(spec):
    from oracle import em_ref
    w, h, fps, balls, frames, seed, clutter, tilt = spec
    scene = synth.JugglingScene(w, h, fps, balls, frames, seed, tilt_deg=tilt)
    meas = synth.make_tags(scene, seed + 1, clutter_per_frame=clutter)
    clip, key = make_clip(meas, w, h, fps, frames)
    clip.build_initial_arcs()
    n3 = len(clip.arcs()[0])
    clip.em_optimize()
    sc = em_ref.Scene(meas, w, h, fps, frame_count=frames)
    sc.build_initial_arcs()
    assert len(sc.arcs) == n3
    sc.em_optimize()
    assert close(clip.state()['camera_tilt'], sc.tilt)
    params, ids, tags = clip.arcs()
    assert len(params) == len(sc.arcs) > 5
    for p, i, t, arc in zip(params, ids, tags, sc.arcs):
        assert i == arc.id and list(t) == arc.tags
        for got, want in zip(p, arc.params()):
            assert close(got, want)
    alive, final = clip.tag_state()
    arc_no = {a: k for k, a in enumerate(sc.arcs)}
    assert list(alive) == sc.alive
    assert [int(final[i]) for i in range(clip.n) if alive[i]] == \
        [arc_no[sc.final_arc[i]] for i in range(sc.n) if sc.alive[i]]


def test_empty_and_tiny_clips():
    from hawkeye_amd.arcs import Clip
    clip = Clip(640, 480, 30.0, 20, [], [], [], [])
    clip.build_initial_arcs()
    clip.em_optimize()
    assert len(clip.arcs()[0]) == 0
    clip = Clip(640, 480, 30.0, 20, [0, 1, 2], [10., 11., 12.],
                [10., 11., 13.], [8., 8., 8.])
    clip.build_initial_arcs()
    clip.em_optimize()
    assert len(clip.arcs()[0]) == 0
    alive, final = clip.tag_state()
    assert (final == -1).all()


def test_device_merge_trials_equal_host_trials():
    """K8: with the self-check on, every trial fit taken from the device is
    recomputed on the host and must match bit for bit (the step fails
    otherwise); the outcome must also equal the run without the check.  The
    noisy 1080p input has thousands of candidate pairs and ~35 mergers."""
    from hawkeye_amd import _native as nat
    scene = synth.JugglingScene(1920, 1080, 60.0, 9, 600, 611)
    meas = synth.make_tags(scene, 612, clutter_per_frame=25.0)
    outcomes = []
    for check in (1, 0):
        nat.check(nat.lib.hk_set_self_check(check))
        try:
            clip, _ = make_clip(meas, 1920, 1080, 60.0, 600)
            clip.build_initial_arcs()
            n3 = len(clip.arcs()[0])
            clip.em_optimize()
            params, ids, tags = clip.arcs()
            outcomes.append((n3, params.tobytes(), ids.tobytes(),
                             [t.tobytes() for t in tags]))
            clip.close()
        finally:
            nat.check(nat.lib.hk_set_self_check(0))
    assert outcomes[0] == outcomes[1]
    assert len(outcomes[0][3]) < outcomes[0][0]        # mergers happened


def test_merge_trials_with_thousands_of_arcs():
    """A long clip: well over a thousand preliminary arcs.  The merge pass
    finds a pair's device trial by bisection in the sorted pair list (memory
    linear in the pairs, not quadratic in the arcs); with the self-check on a
    wrong look-up would fail the step, and the outcome equals the unchecked
    run."""
    from hawkeye_amd import _native as nat
    frames = 16000
    scene = synth.JugglingScene(1920, 1080, 60.0, 9, frames, 811)
    meas = synth.make_tags(scene, 812, clutter_per_frame=10.0)
    outcomes = []
    for check in (1, 0):
        nat.check(nat.lib.hk_set_self_check(check))
        try:
            clip, _ = make_clip(meas, 1920, 1080, 60.0, frames)
            clip.build_initial_arcs()
            n3 = len(clip.arcs()[0])
            clip.em_optimize()
            params, ids, tags = clip.arcs()
            outcomes.append((n3, params.tobytes(), ids.tobytes()))
            clip.close()
        finally:
            nat.check(nat.lib.hk_set_self_check(0))
    print('preliminary arcs:', outcomes[0][0], 'after EM:',
          len(outcomes[0][2]) // 8)
    assert outcomes[0][0] > 1000
    assert outcomes[0] == outcomes[1]


def _result(clip):
    params, ids, tags = clip.arcs()
    alive, final = clip.tag_state()
    return (params.tobytes(), ids.tobytes(), tuple(t.tobytes() for t in tags),
            alive.tobytes(), final.tobytes(), clip.state()['camera_tilt'])


def test_em_batched_across_clips_equals_clip_by_clip(golden_dir):
    """hk_clips_em_optimize: step 4 of several clips in lockstep, every EM
    kernel launched once over the arcs of all of them (BASELINE configs[4]).
    The four golden runs of the reference (different sizes, frame rates,
    numbers of EM rounds), a clip without tags, one whose tags form no arc
    and a duplicate, in one batch: every clip must end bit for bit as it does
    through hk_clip_em_optimize on its own -- and the golden runs are the
    reference's results (test_full_runs_match_reference checks the single
    path against them)."""
    from hawkeye_amd.arcs import Clip

    def build():
        clips = []
        for name in RUNS + [RUNS[1]]:
            with open(os.path.join(golden_dir, 'run_' + name + '.json')) as fh:
                run = json.load(fh)
            clip, _ = make_clip(meas_of(run['meas']), run['width'],
                                run['height'], run['fps'], run['frames'])
            clips.append(clip)
        empty = np.zeros(0)
        clips.insert(2, Clip(640, 480, 30.0, 50, empty.astype(np.int32),
                             empty, empty, empty))
        rng = np.random.RandomState(5)
        f = np.sort(rng.randint(0, 40, 60)).astype(np.int32)
        clips.insert(4, Clip(640, 480, 30.0, 40, f, rng.uniform(0, 640, 60),
                             rng.uniform(0, 480, 60), np.full(60, 8.0)))
        for c in clips:
            c.build_initial_arcs()
        return clips

    alone = build()
    for c in alone:
        c.em_optimize()
    want = [_result(c) for c in alone]
    assert sum(len(w[2]) for w in want) > 100
    assert len(want[2][2]) == 0 and len(want[4][2]) == 0

    batch = build()
    Clip.em_optimize_many(batch)
    got = [_result(c) for c in batch]
    assert got == want

    # any subset, any order
    batch = build()
    order = [6, 0, 4, 3]
    Clip.em_optimize_many([batch[i] for i in order])
    assert [_result(batch[i]) for i in order] == [want[i] for i in order]

    # a clip that ran single operations keeps its own batch and is refused
    # by the batched entry point, with a message
    from hawkeye_amd import _native as nat
    batch = build()
    batch[0].calculate_weights()
    with pytest.raises(nat.NativeError, match='single EM operations'):
        Clip.em_optimize_many(batch[:2])
    batch[0].em_optimize()      # ... and carries on by itself
    for c in alone + batch:
        c.close()
