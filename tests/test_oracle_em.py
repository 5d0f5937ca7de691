"""
The steps 3-4 oracle (oracle/em_ref.py) against golden vectors produced by
the reference itself (tests/golden/make_golden.py).  CPU only.

Tolerances: the reference sums in Python-set order (address dependent), the
oracle in tag-index order, so values agree to rounding, not bits.  REL is far
inside north_star's 1e-3 bound on final arc parameters.
"""
import json
import math
import os

import pytest

from oracle import em_ref

REL = 1e-9
# full runs chain dozens of ill-conditioned fits (determinants cancel ~7
# digits), so summation-order noise grows to ~1e-9; 1e-6 is still 1000x
# inside north_star's 1e-3
REL_RUN = 1e-6
RUNS = ['clean_720p30_5ball', 'noisy_1080p60_9ball', 'tilt3_720p30_5ball',
        'lowfps_480p24_3ball']


def close(a, b, rel=REL, abs_=1e-9):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


def meas_of(d):
    return {int(f): [tuple(t) for t in lst] for f, lst in d.items()}


@pytest.fixture(scope='module')
def unit(golden_dir):
    return load(golden_dir, 'unit_vectors.json')


def test_geometry(unit):
    for g in unit['geometry']:
        sc = em_ref.Scene({}, g['notes']['frame_width'],
                          g['notes']['frame_height'], 30.0,
                          frame_count=g['notes']['frame_count'])
        sc.tilt = g['tilt']
        arc = em_ref.Arc()
        arc.a, arc.b, arc.c, arc.e, arc.f_peak = g['params']
        for f, want in zip(g['frames'], g['positions']):
            got = sc.position(arc, f)
            assert got[0] == want[0] and got[1] == want[1]
        assert list(sc.frame_range(arc)) == g['frame_range']
        sc.frame, sc.x, sc.y = [g['tag'][0]], [g['tag'][1]], [g['tag'][2]]
        assert sc.dist(arc, 0) == g['tag_distance']


def test_fit_arcs(unit):
    for case in unit['fit_arcs']:
        sc = em_ref.Scene({}, 1280, 720, 30.0, frame_count=200)
        sc.tilt = case['tilt']
        tags = case['tags']
        sc.frame = [t[0] for t in tags]
        sc.x = [t[1] for t in tags]
        sc.y = [t[2] for t in tags]
        wts = [t[3] for t in tags]
        arc = em_ref.Arc()
        arc.a, arc.b, arc.c, arc.e, arc.f_peak = case['start']
        arc.tags = list(range(len(tags)))
        sc.fit(arc, wts.__getitem__)
        for got, want in zip(arc.params(), case['result']):
            assert close(got, want), (got, want)


def test_fit_arcs_unit_weights_and_skip_paths(golden_dir):
    """The reference's fit_arcs() with unit weights, including every way it
    leaves an arc untouched (fit_unit_vectors.json)."""
    cases = load(golden_dir, 'fit_unit_vectors.json')['fit_arcs_unit_weights']
    assert sum(c['skipped'] for c in cases) >= 5
    for case in cases:
        sc = em_ref.Scene({}, 1280, 720, 30.0, frame_count=200)
        sc.tilt = case['tilt']
        tags = case['tags']
        sc.frame = [t[0] for t in tags]
        sc.x = [t[1] for t in tags]
        sc.y = [t[2] for t in tags]
        arc = em_ref.Arc()
        arc.a, arc.b, arc.c, arc.e, arc.f_peak = case['start']
        arc.tags = list(range(len(tags)))
        sc.fit(arc, lambda i: 1.0)
        if case['skipped']:
            assert list(arc.params()) == case['start'], case['name']
        else:
            for got, want in zip(arc.params(), case['result']):
                assert close(got, want), (case['name'], got, want)


def test_estep_tilt_eval(unit):
    u = unit['estep']
    sc = em_ref.Scene(meas_of(u['meas']), u['width'], u['height'], u['fps'],
                      frame_count=u['frames'])
    sc.build_initial_arcs()
    assert close(sc.g, u['g_px_per_frame_sq'])
    index = {k: i for i, k in enumerate(sc.key)}
    assert len(sc.arcs) == len(u['arcs_after_step3'])
    for arc, want in zip(sc.arcs, u['arcs_after_step3']):
        assert [list(sc.key[i]) for i in arc.tags] == want['tags']
        for k in ('a', 'b', 'c', 'e', 'f_peak'):
            assert close(getattr(arc, k), want[k])
    sc.estimate_tilt()
    assert close(sc.tilt, u['camera_tilt'])
    sc.fit_all()
    for arc, want in zip(sc.arcs, u['arcs_after_fit']):
        for k in ('a', 'b', 'c', 'e', 'f_peak'):
            assert close(getattr(arc, k), want[k])
    sc.e_step()
    arc_no = {a: k for k, a in enumerate(sc.arcs)}
    got = sorted([sc.key[i][0], sc.key[i][1], arc_no[a], w]
                 for i in range(sc.n) for a, w in sc.w[i].items())
    want = sorted(u['weights'])
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:3] == w[:3] and close(g[3], w[3])
    for arc, members in zip(sc.arcs, u['members']):
        assert [list(sc.key[i]) for i in arc.tags] == members
    for arc, codes in zip(sc.arcs, u['eval_codes']):
        assert [sc.eval_arc(arc, False), sc.eval_arc(arc, True)] == codes

    a0 = sc.arcs[0]
    bad = dict(u['bad_arc_codes'])
    nan_arc = em_ref.Arc(); nan_arc.a = float('nan')
    assert sc.eval_arc(nan_arc) == bad['nan'] == 1
    empty = em_ref.Arc(); empty.e = 0.5
    assert sc.eval_arc(empty) == bad['empty'] == 5
    t = a0.clone(); t.e = -a0.e
    assert sc.eval_arc(t) == bad['neg_accel'] == 4
    t = a0.clone(); t.e = a0.e * 0.5
    assert sc.eval_arc(t) == bad['half_accel'] == 4
    t = a0.clone(); t.a = a0.a + 500.0
    assert sc.eval_arc(t) == bad['shifted'] == 2
    t = a0.clone(); t.f_peak = a0.f_peak + 500.0
    assert sc.eval_arc(t, requirepeak=True, checkaccel=False) \
        == bad['no_peak'] == 3


@pytest.mark.parametrize('name', RUNS)
def test_full_run(golden_dir, name):
    run = load(golden_dir, 'run_{}.json'.format(name))
    sc = em_ref.Scene(meas_of(run['meas']), run['width'], run['height'],
                      run['fps'], frame_count=run['frames'])
    sc.build_initial_arcs()
    s3 = run['step3']
    assert close(sc.g, s3['g_px_per_frame_sq'])
    assert close(sc.cm_per_pixel, s3['cm_per_pixel'])
    assert len(sc.arcs) == len(s3['arcs'])
    for arc, want in zip(sc.arcs, s3['arcs']):
        assert arc.id == want['id']
        assert [list(sc.key[i]) for i in arc.tags] == want['tags']
        for k in ('a', 'b', 'c', 'e', 'f_peak'):
            assert close(getattr(arc, k), want[k], rel=REL_RUN)

    sc.em_optimize()
    s4 = run['step4']
    assert close(sc.tilt, s4['camera_tilt'], rel=REL_RUN, abs_=1e-9)
    assert len(sc.arcs) == len(s4['arcs'])
    for arc, want in zip(sc.arcs, s4['arcs']):
        assert arc.id == want['id']
        for k in ('a', 'b', 'c', 'e', 'f_peak'):
            assert close(getattr(arc, k), want[k], rel=REL_RUN), (k, arc.id)
    arc_no = {a: k for k, a in enumerate(sc.arcs)}
    got = [[sc.key[i][0], sc.key[i][1], arc_no[sc.final_arc[i]]]
           for i in range(sc.n) if sc.alive[i]]
    assert got == s4['assign']
    assert len(got) == s4['tags_remaining']
