"""
GPU: the array-level clip driver (hawkeye_amd.batch) -- one clip, a pipelined
batch of clips, and agreement with the object-level VideoScanner.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def small_spec(seed, frames=150, balls=3):
    from hawkeye_amd import synth
    spec = dict(synth.CLIPS['B_720p30_5ball'])
    spec.update(width=320, height=240, frames=frames, seed=seed, balls=balls,
                ball_r=4.0)
    return spec


def render(spec):
    from hawkeye_amd import synth
    return synth.ClipRenderer(spec, device=0).render(0, spec['frames'])


def same_result(a, b):
    assert len(a.events) == len(b.events)
    for p, q in zip(a.events, b.events):
        assert np.array_equal(p.view(np.uint32), q.view(np.uint32))
    assert np.array_equal(a.arc_params, b.arc_params)
    assert np.array_equal(a.arc_ids, b.arc_ids)
    assert all(np.array_equal(m, n)
               for m, n in zip(a.arc_members, b.arc_members))
    assert np.array_equal(a.alive, b.alive)
    assert np.array_equal(a.final_arc, b.final_arc)
    assert a.camera_tilt == b.camera_tilt


@pytest.mark.parametrize('lanes', [1, 2, 3])
def test_pipelined_batch_equals_clip_by_clip(lanes, monkeypatch):
    from hawkeye_amd.batch import ClipScanner
    monkeypatch.setenv('HK_DETECT_LANES', str(lanes))
    specs = [small_spec(4100 + i, balls=3 + i % 3) for i in range(5)]
    clips = [render(s) for s in specs]
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, specs[0]['fps'], device=0, chunk=64)
    one_by_one = [sc.scan(c) for c in clips]
    assert any(len(r.arc_params) for r in one_by_one)
    batched = list(sc.scan_many(iter(clips)))
    assert len(batched) == len(clips)
    for a, b in zip(one_by_one, batched):
        same_result(a, b)
    # again, to catch state leaking from one batch into the next
    for a, b in zip(one_by_one, sc.scan_many(clips)):
        same_result(a, b)
    sc.close()


@pytest.mark.parametrize('lanes', [1, 2])
def test_ragged_and_empty_clips_with_and_without_lookahead(lanes, monkeypatch):
    """Clip lengths that are no multiple of the chunk (cut into equal parts),
    shorter than a chunk, and zero: scan_many with the chunk pipeline kept
    full across clips (the default) and drained at every clip's end gives
    what scan() gives clip by clip."""
    from hawkeye_amd.batch import ClipScanner
    monkeypatch.setenv('HK_DETECT_LANES', str(lanes))
    lengths = [70, 33, 0, 129, 1, 64]
    clips = [render(small_spec(4250 + i, frames=max(n, 1)))[:n]
             for i, n in enumerate(lengths)]
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, 30.0, device=0, chunk=32)
    assert [len(list(sc._chunks_of(c))) for c in clips] == [3, 2, 0, 5, 1, 2]
    assert [int(p.shape[0]) for p in sc._chunks_of(clips[3])] == \
        [28, 28, 28, 28, 17]
    one_by_one = [sc.scan(c) for c in clips]
    assert [len(r.events) for r in one_by_one] == lengths
    for lookahead in (True, False):
        batched = list(sc.scan_many(iter(clips), lookahead=lookahead))
        assert len(batched) == len(clips)
        for a, b in zip(one_by_one, batched):
            same_result(a, b)
    sc.close()


@pytest.mark.parametrize('lanes', [1, 2])
def test_batch_stops_cleanly_when_the_consumer_stops(lanes, monkeypatch):
    from hawkeye_amd.batch import ClipScanner
    monkeypatch.setenv('HK_DETECT_LANES', str(lanes))
    clips = [render(small_spec(4200 + i, frames=80)) for i in range(4)]
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, 30.0, device=0, chunk=32)
    gen = sc.scan_many(clips)
    first = next(gen)
    gen.close()                     # abandons the remaining clips
    again = sc.scan(clips[0])
    same_result(first, again)
    sc.close()


@pytest.mark.parametrize('lanes', [1, 2])
def test_errors_in_a_stage_reach_the_caller(lanes, monkeypatch):
    from hawkeye_amd.batch import ClipScanner
    monkeypatch.setenv('HK_DETECT_LANES', str(lanes))
    good = render(small_spec(4300, frames=60))
    bad = torch.zeros((10, 100, 100, 3), dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, 30.0, device=0, chunk=32)
    with pytest.raises(ValueError):
        list(sc.scan_many([good, bad, good]))
    same_result(sc.scan(good), sc.scan(good))
    sc.close()


def test_matches_videoscanner_notes():
    """Same events and arcs as the reference-shaped VideoScanner produces."""
    from hawkeye_amd import frames as fsrc
    from hawkeye_amd.batch import ClipScanner
    from hawkeye_amd.tracker import VideoScanner
    spec = small_spec(4400, frames=200, balls=3)
    clip = render(spec)
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, spec['fps'], device=0, chunk=64)
    res = sc.scan(clip)
    sc.close()

    path = '/virtual/batch_vs_scanner.mov'
    fsrc.register_source(path, lambda: fsrc.ArraySource(clip, spec['fps']))
    try:
        vs = VideoScanner(path)
        vs.process(steps=(1, 4))
    finally:
        fsrc.unregister_source(path)
    notes = vs.notes
    # events of step 2 that survived clean_notes, in order
    kept = [(t.frame, t.x, t.y, t.radius)
            for f in range(notes['frame_count']) for t in notes['meas'][f]]
    mine = [(int(f), x, y, r) for f, x, y, r, a in
            zip(res.frame, res.x, res.y, res.radius, res.alive) if a]
    assert kept == mine
    arcs = sorted(notes['arcs'], key=lambda a: a.f_peak)
    assert len(arcs) == len(res.arc_params)
    for a, p in zip(arcs, res.arc_params):
        assert (a.a, a.b, a.c, a.e, a.f_peak) == tuple(p)
    assert notes['camera_tilt'] == res.camera_tilt


def test_pinned_host_source_streams_the_same_frames():
    """Frames fed from page-locked host memory (double-buffered upload on a
    copy stream) give the events of the device-resident clip, for chunk
    counts that do and do not divide the clip."""
    from hawkeye_amd.batch import ClipScanner
    from hawkeye_amd.frames import PinnedHostSource
    spec = small_spec(4500, frames=230, balls=4)
    clip = render(spec)
    torch.cuda.synchronize()
    for chunk in (32, 64, 230):
        sc = ClipScanner(320, 240, spec['fps'], device=0, chunk=chunk)
        want = sc.detect(clip)
        host = PinnedHostSource.alloc(spec['frames'], 240, 320)
        host.copy_(clip)
        torch.cuda.synchronize()
        src = PinnedHostSource(host, spec['fps'], device=0, chunk=chunk)
        got = sc.detect(src)
        src.close()
        # a pageable NumPy clip is accepted too (pinned once up front)
        src = PinnedHostSource(clip.cpu().numpy(), spec['fps'], device=0,
                               chunk=chunk)
        got2 = sc.detect(src)
        src.close()
        sc.close()
        assert len(got) == len(want) == len(got2) == spec['frames']
        for a, b, c in zip(want, got, got2):
            assert np.array_equal(a, b) and np.array_equal(a, c)


def test_tile_path_equals_byte_path(monkeypatch):
    from hawkeye_amd import synth
    """Step 2 with the MOG kernel's masks handed to the blob detector as
    1-bit tiles (the default where frames are made of 32x4 tiles) and as byte
    images gives the same keypoints, bit for bit; so does a chunk length that
    leaves a ragged last chunk."""
    from hawkeye_amd.detect import EventDetector, hawkeye_blob_params
    spec = dict(width=640, height=360, fps=60.0, balls=7, frames=150,
                ball_r=6.0, sigma=3.0, drift=6.0, drift_hz=0.3,
                distractors=12, seed=4242)
    frames = synth.ClipRenderer(spec).render(0, spec['frames'])

    def run(tiles, chunk):
        monkeypatch.setenv('HK_DETECT_TILES', '1' if tiles else '0')
        det = EventDetector(spec['width'], spec['height'],
                            hawkeye_blob_params(spec['height']), chunk=chunk)
        assert det.use_tiles == tiles
        out = []
        for s in range(0, spec['frames'], chunk):
            out += det.submit(frames[s:s + chunk])
        out += det.flush()
        det.close()
        return out

    want = run(False, 64)
    assert sum(len(k) for k in want) > 500
    for chunk in (64, 37, 1):
        got = run(True, chunk)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_frames_produced_just_before_the_scan_are_waited_for():
    """Clips rendered lazily on the caller's stream, right before they are
    handed to scan_many (no synchronisation by the caller), give the events
    of clips rendered and synchronised beforehand."""
    from hawkeye_amd import synth
    from hawkeye_amd.batch import ClipScanner
    specs = [dict(width=1280, height=720, fps=60.0, balls=3 + i, frames=200,
                  ball_r=8.0, sigma=3.0, drift=0.0, drift_hz=0.0,
                  distractors=4, seed=7100 + i) for i in range(6)]
    sc = ClipScanner(1280, 720, 60.0)
    ready = [synth.ClipRenderer(s).render(0, s['frames']) for s in specs]
    torch.cuda.synchronize()
    want = [sc.scan(f) for f in ready]
    del ready
    torch.cuda.synchronize()

    def lazily():
        for s in specs:
            yield synth.ClipRenderer(s).render(0, s['frames'])
    got = list(sc.scan_many(lazily()))
    sc.close()
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert len(a.events) == len(b.events)
        for x, y in zip(a.events, b.events):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        assert np.array_equal(a.arc_params, b.arc_params)


def test_clips_with_tags_but_no_arcs_between_normal_ones():
    """A clip whose events form no arc (static scene plus sensor noise blobs,
    or a handful of frames) goes through step 4 without ever waiting for its
    EM stream: its pooled device workspace must not change hands while the
    tag uploads are still queued (hk_em.hip, wait_for_uploads).  Many short
    no-arc clips between normal ones, every builder thread on its own
    stream: the normal clips' results must not move."""
    from hawkeye_amd.batch import ClipScanner
    normal = [render(small_spec(4400 + i, frames=120, balls=3 + i % 2))
              for i in range(3)]
    # a few frames of a ball clip: events (after the first frame) but far too
    # few for an arc
    stub = [render(small_spec(4450 + i, frames=12))[:3 + i] for i in range(4)]
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, 30.0, device=0, chunk=32)
    want_normal = [sc.scan(c) for c in normal]
    want_stub = [sc.scan(c) for c in stub]
    assert all(len(r.arc_params) > 0 for r in want_normal)
    assert all(len(r.arc_params) == 0 for r in want_stub)
    assert any(len(r.frame) > 0 for r in want_stub)
    order, want = [], []
    for rep in range(6):
        for i in range(3):
            order += [normal[i], stub[(rep + i) % 4], stub[(rep + i + 1) % 4]]
            want += [want_normal[i], want_stub[(rep + i) % 4],
                     want_stub[(rep + i + 1) % 4]]
    got = list(sc.scan_many(iter(order), count=len(order)))
    assert len(got) == len(want)
    for a, b in zip(want, got):
        same_result(a, b)
    sc.close()


def test_an_error_of_the_clip_iterator_loses_no_finished_clip():
    """The caller's iterator raises after two clips: both were submitted in
    full, so both results reach the caller before the error does."""
    from hawkeye_amd.batch import ClipScanner
    clips = [render(small_spec(4500 + i, frames=70)) for i in range(2)]
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, 30.0, device=0, chunk=32)
    want = [sc.scan(c) for c in clips]

    def source():
        yield clips[0]
        yield clips[1]
        raise KeyError('no more video')

    got = []
    with pytest.raises(KeyError):
        for res in sc.scan_many(source()):
            got.append(res)
    assert len(got) == 2
    for a, b in zip(want, got):
        same_result(a, b)
    sc.close()


@pytest.mark.parametrize('mode', ['em_thread', 'stream_per_builder'])
def test_step4_modes_of_the_pipeline_give_the_same_results(mode, monkeypatch):
    """scan_many with step 4 on one lockstep thread (hk_clips_em_optimize over
    the clips that are waiting) and with a stream per builder thread, against
    the default (builders run step 4 of their own clip on the shared EM
    stream) and against clip-by-clip scanning."""
    from hawkeye_amd.batch import ClipScanner
    specs = [small_spec(4600 + i, balls=3 + i % 3, frames=120 + 10 * i)
             for i in range(7)]
    clips = [render(s) for s in specs]
    torch.cuda.synchronize()
    sc = ClipScanner(320, 240, specs[0]['fps'], device=0, chunk=64)
    want = [sc.scan(c) for c in clips]
    if mode == 'em_thread':
        monkeypatch.setenv('HK_EM_THREAD', '1')
        monkeypatch.setenv('HK_EM_BATCH', '4')
    else:
        monkeypatch.setenv('HK_EM_SHARED_STREAM', '0')
    for _ in range(2):
        got = list(sc.scan_many(iter(clips), builders=4))
        assert len(got) == len(want)
        for a, b in zip(want, got):
            same_result(a, b)
    if mode == 'em_thread':
        assert all(r.stage_seconds['step4_batch'] >= 1 for r in got)
    sc.close()
