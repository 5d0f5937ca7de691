"""
GPU: the VideoScanner drop-in end to end (steps 1-4) on small synthetic
clips, every stage checked against the oracles on the same frames:
events bit for bit, arcs to 1e-6, plus the caller contract (callbacks,
abort, notes files, ScannerException).
"""
import os

import numpy as np
import pytest

from hawkeye_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

SPEC = dict(width=320, height=240, fps=30.0, balls=3, frames=150, ball_r=4.0,
            sigma=2.0, drift=0.0, drift_hz=0.0, distractors=3, seed=41)
SPEC_HI = dict(width=640, height=480, fps=60.0, balls=4, frames=120,
               ball_r=6.0, sigma=3.0, drift=4.0, drift_hz=0.3, distractors=4,
               seed=42)


def close(a, b, rel=1e-6, abs_=1e-9):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


@pytest.fixture
def clip_path(tmp_path):
    from hawkeye_amd import frames
    made = []

    def make(spec, name='clip.mov'):
        path = str(tmp_path / name)
        frames.register_source(path, lambda: frames.SyntheticSource(spec))
        made.append(path)
        return path
    yield make
    for p in made:
        frames.unregister_source(p)


def oracle_events(spec):
    """Step 2 by the C oracle on the same rendered frames (copied to host)."""
    from oracle import detect_ref as D
    r = synth.ClipRenderer(spec)
    frames = r.render(0, spec['frames']).cpu().numpy()
    mog = D.Mog(spec['width'], spec['height'])
    p = D.hawkeye_params(spec['height'])
    return [D.blob_detect(mog.apply(f), p) for f in frames]


@pytest.mark.parametrize('spec', [SPEC, SPEC_HI])
def test_steps_1_to_4_against_oracles(clip_path, spec):
    from hawkeye_amd.tracker import VideoScanner
    from oracle import em_ref
    path = clip_path(spec)
    seen = []
    sc = VideoScanner(path)
    sc.process(steps=(1, 2), callback=lambda *a: seen.append(a))
    notes = sc.notes
    assert notes['step'] == 2
    assert notes['frame_count'] == spec['frames']
    assert (notes['frame_width'], notes['frame_height']) == (spec['width'],
                                                             spec['height'])
    assert seen == [(f + 1, spec['frames']) for f in range(spec['frames'])]
    assert sorted(notes['meas']) == list(range(spec['frames']))
    assert notes['camera_tilt'] == 0.0
    sp = notes['scanner_params']
    assert sp['max_distance_pixels'] == 5 * spec['height'] / 480
    assert sp['min_tags_per_arc'] == 10
    assert sp['radius_window'] == (0.65 if spec['height'] >= 480 else 0.75)

    want = oracle_events(spec)
    total = 0
    for f in range(spec['frames']):
        got = [(t.x, t.y, t.radius) for t in notes['meas'][f]]
        exp = [(float(k[0]), float(k[1]), float(k[2])) for k in want[f]]
        assert got == exp, 'frame {}'.format(f)
        assert all(t.frame == f and t.total_weight == 1.0 and t.arc is None
                   for t in notes['meas'][f])
        total += len(got)
    assert total > spec['frames']           # balls were seen

    meas = {f: [(t.x, t.y, t.radius) for t in notes['meas'][f]]
            for f in range(spec['frames'])}
    ref = em_ref.Scene(meas, spec['width'], spec['height'], spec['fps'],
                       frame_count=spec['frames'])
    ref.build_initial_arcs()

    calls = []
    sc.process(steps=(3, 3), callback=lambda *a: calls.append(a))
    assert len(notes['arcs']) == len(ref.arcs) > 2
    assert close(notes['g_px_per_frame_sq'], ref.g)
    assert close(notes['cm_per_pixel'], ref.cm_per_pixel)
    for arc, want_arc in zip(notes['arcs'], ref.arcs):
        assert arc.id_ == want_arc.id
        assert len(arc.tags) == len(want_arc.tags)
        for t in arc.tags:
            assert t.weight == {arc: 1.0}

    ref.em_optimize()
    sc.process(steps=(4, 4))
    assert notes['step'] == 4
    assert close(notes['camera_tilt'], ref.tilt)
    assert len(notes['arcs']) == len(ref.arcs)
    for arc, want_arc in zip(notes['arcs'], ref.arcs):
        assert arc.id_ == want_arc.id
        for k, v in zip(('a', 'b', 'c', 'e', 'f_peak'), want_arc.params()):
            assert close(getattr(arc, k), v)
        assert all(t.arc is arc for t in arc.tags)
        assert len(arc.tags) == len(want_arc.tags)
    left = sum(len(notes['meas'][f]) for f in range(spec['frames']))
    assert left == sum(ref.alive) == sum(len(a.tags) for a in notes['arcs'])
    for f in range(spec['frames']):
        for t in notes['meas'][f]:
            assert not hasattr(t, 'weight') and not hasattr(t, 'done')

    # the scene's true flights are among the arcs found
    scene = synth.JugglingScene(spec['width'], spec['height'], spec['fps'],
                                spec['balls'], spec['frames'], spec['seed'])
    peaks = [a.f_peak for a in notes['arcs']]
    inside = [fl for fl in scene.flights
              if fl.f0 >= 5 and fl.f1 <= spec['frames'] - 5]
    hit = sum(any(abs(fl.f_peak - p) < 1.0 for p in peaks) for fl in inside)
    if spec is SPEC_HI:     # the low-res clip's 14-frame flights rarely reach
        assert hit >= 0.8 * len(inside)     # min_tags_per_arc = 10
    else:
        assert hit >= 1
    for a in notes['arcs']:
        assert close(2 * a.e, scene.g, rel=0.15)


def test_notes_files_and_resume(clip_path, tmp_path):
    from hawkeye_amd.tracker import VideoScanner
    path = clip_path(SPEC)
    a = VideoScanner(path)
    a.process(steps=(1, 3), writenotes=True, notesdir='__Hawkeye__')
    f3 = tmp_path / '__Hawkeye__' / 'clip_notes3.pkl'
    assert f3.exists()
    b = VideoScanner(path)
    b.process(steps=(4, 4), readnotes=True, writenotes=True,
              notesdir='__Hawkeye__')
    assert (tmp_path / '__Hawkeye__' / 'clip_notes4.pkl').exists()
    a.process(steps=(4, 4))
    assert len(a.notes['arcs']) == len(b.notes['arcs']) > 0
    for p, q in zip(a.notes['arcs'], b.notes['arcs']):
        assert (p.a, p.b, p.c, p.e, p.f_peak, p.id_) == \
            (q.a, q.b, q.c, q.e, q.f_peak, q.id_)
    back = VideoScanner.read_notes(str(tmp_path / '__Hawkeye__' /
                                       'clip_notes4.pkl'))
    assert type(back['arcs'][0]).__module__ == 'hawkeye.types'
    assert back['step'] == 4 and back['version'] == 4


def test_scan_video_coordinates(clip_path):
    """A half-size scan video: events come back in source coordinates
    (reference tracker.py:335-342, 413-417)."""
    from hawkeye_amd.tracker import VideoScanner
    src = dict(SPEC_HI, frames=40)
    small = dict(SPEC, frames=40, fps=60.0)
    p_src = clip_path(src, 'src.mov')
    p_scan = clip_path(small, 'scan.mov')
    sc = VideoScanner(p_src, scanvideo=p_scan)
    sc.process(steps=(1, 2))
    direct = VideoScanner(p_scan)
    direct.process(steps=(1, 2))
    n = 0
    for f in range(40):
        for t, u in zip(sc.notes['meas'][f], direct.notes['meas'][f]):
            assert t.x == (640 - 480 * (320 / 240)) / 2 + u.x * 2.0
            assert t.y == u.y * 2.0 and t.radius == u.radius * 2.0
            n += 1
    assert n > 20
    assert sc.notes['scanner_params']['radius_window'] == 0.75
    assert sc.notes['scanner_params']['max_distance_pixels'] == 5.0


def test_abort_and_errors(clip_path, tmp_path):
    from hawkeye_amd.tracker import VideoScanner, ScannerException
    path = clip_path(SPEC)

    class Abort(Exception):
        pass

    def cb(*args):
        if args and args[0] == 37:
            raise Abort()
    sc = VideoScanner(path)
    with pytest.raises(Abort):
        sc.process(steps=(1, 2), callback=cb)
    with pytest.raises(ScannerException, match='Error opening video file'):
        VideoScanner(str(tmp_path / 'missing.mov')).process(steps=(1, 1))
    ok = VideoScanner(path)
    ok.process(steps=(1, 4))
    assert len(ok.notes['arcs']) > 0


def test_all_six_steps_and_the_notes_file(clip_path, tmp_path):
    """process(steps=(1, 6)) end to end, the way worker.py calls it
    (reference worker.py:757-759): step 5 without a detector leaves body
    positions to step 6's estimate; a plugged-in detector is used; the final
    notes file is `<base>_notes.pkl` and unpickles with hawkeye.types."""
    import pickle
    from hawkeye_amd.tracker import VideoScanner
    path = clip_path(SPEC, 'six.mov')
    calls = []
    sc = VideoScanner(path)
    sc.process(steps=(1, 6), writenotes=True, notesdir=str(tmp_path / 'n'),
               callback=lambda *a: calls.append(a), verbosity=0)
    n = sc.notes
    assert n['step'] == 6 and n['runs'] == len(n['run']) >= 1
    assert all(b[4] is False for b in n['body'].values())
    arcs = n['arcs']
    assert arcs and all(a.run_id is not None and a.throw_id is not None and
                        a.hand_throw in ('left', 'right') and
                        a.hand_catch in ('left', 'right') and
                        a.f_throw < a.f_peak < a.f_catch for a in arcs)
    for info in n['run']:
        assert info['throws'] == len(info['throw']) >= 2
        assert 1 <= info['balls'] <= info['throws']
        first, last = info['frame range']
        assert all(f in n['origin'] or not (first <= f <= last)
                   for f in range(n['frame_count']))
    assert () in calls                          # callback() once per run
    saved = tmp_path / 'n' / 'six_notes.pkl'
    assert saved.exists()
    with open(saved, 'rb') as fh:
        back = pickle.load(fh)
    assert type(back['arcs'][0]).__module__ == 'hawkeye.types'
    assert [a.throw_id for a in back['arcs']] == [a.throw_id for a in arcs]
    assert back['arcs'][0].tags and \
        all(t.arc is back['arcs'][0] for t in back['arcs'][0].tags)

    # with a body detector plugged into step 5
    w, h = SPEC['width'], SPEC['height']
    boxes = {f: (0.4 * w, 0.62 * h, 0.2 * w, 0.35 * h, True)
             for f in range(0, SPEC['frames'], 2)}
    sc2 = VideoScanner(path, body_detector=lambda scanner: boxes)
    sc2.process(steps=(1, 6))
    assert sum(1 for b in sc2.notes['body'].values() if b[4]) == len(boxes)
    assert sc2.notes['runs'] == len(sc2.notes['run'])
    assert sc2.notes['step'] == 6
