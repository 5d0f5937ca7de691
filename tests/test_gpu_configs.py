"""
GPU: BASELINE.json's configurations at their own sizes --
B 720p30 5-ball, D 4K 240 fps (time-segmented layout), E 64 clips 1080p --
against the C oracle on the same rendered frames (step 2, bit for bit, on as
many frames as the oracle does in seconds) and through size-independent
properties on the whole clips: batch = clip by clip, 1-bit tile path = byte
path, the recovered arcs are the rendered flights.  Reference loop covered:
tracker.py:382-433.  Configuration C (the one bench.py times) is tested the
same way, plus a window deep inside the clip with the oracle continuing from
the device's exported model; configs[0] needs a video file and cv2.
"""
import os

import numpy as np
import pytest

from hawkeye_amd import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def oracle_step2(host_frames, width, height):
    """(keypoints per frame, final MOG model) of the C oracle."""
    from oracle import detect_ref as D
    mog = D.Mog(width, height)
    bp = D.hawkeye_params(height)
    kps = [D.blob_detect(mog.apply(f), bp) for f in host_frames]
    return kps, mog.model()


def same_keypoints(got, want):
    assert len(got) == len(want)
    for f, (a, b) in enumerate(zip(got, want)):
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        assert a.shape == b.shape, 'frame {}: {} keypoints, oracle {}'.format(
            f, len(a), len(b))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
            'frame {}'.format(f)


def detect(frames, width, height, chunk, tiles=True, monkeypatch=None):
    """Step 2 through EventDetector; returns (keypoints, detector)."""
    from hawkeye_amd.detect import EventDetector, hawkeye_blob_params
    if monkeypatch is not None:
        monkeypatch.setenv('HK_DETECT_TILES', '1' if tiles else '0')
    det = EventDetector(width, height, hawkeye_blob_params(height),
                        chunk=chunk)
    out = []
    for s in range(0, frames.shape[0], chunk):
        out += det.submit(frames[s:s + chunk])
    out += det.flush()
    return out, det


def flights_recovered(spec, res, tol_px):
    """Fraction of the scene's complete flights that have an arc whose peak
    (frame, x, y) lies within tol of the flight's."""
    scene = synth.JugglingScene(spec['width'], spec['height'], spec['fps'],
                                spec['balls'], spec['frames'], spec['seed'])
    arcs = np.asarray(res.arc_params)           # a b c e f_peak
    found = total = 0
    for fl in scene.flights:
        if fl.f0 < 5 or fl.f1 > spec['frames'] - 5:
            continue                            # cut off by the clip's ends
        total += 1
        xp, yp = fl.position(fl.f_peak)
        if len(arcs) and np.any((np.abs(arcs[:, 4] - fl.f_peak) < 2.0) &
                                (np.abs(arcs[:, 0] - xp) < tol_px) &
                                (np.abs(arcs[:, 2] - yp) < tol_px)):
            found += 1
    return found, total, scene.g


# ---------------------------------------------------------------------------
# B: 1280x720, 30 fps, 5 balls, 900 frames
# ---------------------------------------------------------------------------

def test_config_B_720p30_5ball():
    from hawkeye_amd.batch import ClipScanner
    spec = synth.CLIPS['B_720p30_5ball']
    w, h = spec['width'], spec['height']
    clip = synth.ClipRenderer(spec).render(0, spec['frames'])
    torch.cuda.synchronize()
    sc = ClipScanner(w, h, spec['fps'])
    res = sc.scan(clip)
    assert len(res.events) == spec['frames']

    # step 2 against the oracle, bit for bit, on the first 64 frames
    want, _ = oracle_step2(clip[:64].cpu().numpy(), w, h)
    same_keypoints(res.events[:64], want)
    assert sum(len(k) for k in want) > 100

    # the whole clip: the pipelined batch gives the same result ...
    again = list(sc.scan_many([clip, clip]))
    sc.close()
    for other in again:
        same_keypoints(other.events, res.events)
        assert np.array_equal(other.arc_params, res.arc_params)
    # ... and the arcs are the flights that were rendered (ideal parabolas,
    # ball radius 8 px): every complete flight is found, with the scene's g
    found, total, g = flights_recovered(spec, res, tol_px=4.0)
    print('config B: {} events, {} arcs, {}/{} flights recovered'.format(
        sum(len(k) for k in res.events), len(res.arc_params), found, total))
    assert total > 100 and found >= 0.97 * total
    assert abs(res.g_px_per_frame_sq / g - 1.0) < 0.02
    assert abs(res.camera_tilt) < 0.01


# ---------------------------------------------------------------------------
# C: 1920x1080, 60 fps, 9 balls, noisy drifting background, 1800 frames --
# the workload bench.py's headline number is measured on
# ---------------------------------------------------------------------------

# bench.py's own report for this clip (its renderer is deterministic)
CONFIG_C_EVENTS, CONFIG_C_ARCS = 49861, 174


def test_config_C_1080p60_9ball_noisy():
    from hawkeye_amd.batch import ClipScanner
    from hawkeye_amd.detect import EventDetector, hawkeye_blob_params
    from oracle import detect_ref as D
    spec = synth.CLIPS['C_1080p60_9ball_noisy']
    w, h = spec['width'], spec['height']
    clip = synth.ClipRenderer(spec).render(0, spec['frames'])
    torch.cuda.synchronize()
    sc = ClipScanner(w, h, spec['fps'])
    res = sc.scan(clip)
    assert len(res.events) == spec['frames'] == 1800

    # step 2 against the oracle, bit for bit: frames 0-255 (the alpha ramp of
    # the first 200 frames and the first chunk borders; the oracle threaded
    # over rows and frames gives the single-thread events, see
    # tests/test_oracle_detect.py) ...
    bp = D.hawkeye_params(h)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    want = D.detect_clip(clip[:256].cpu().numpy(), bp, threads=threads)
    same_keypoints(res.events[:256], want)
    assert sum(len(k) for k in want) > 5000

    # ... and a window past frame 1000, where most tiles have left the
    # kernel's unit path: a detector run up to frame 1024 hands its model to
    # the oracle, which continues for 32 frames
    det = EventDetector(w, h, hawkeye_blob_params(h), chunk=128)
    head = []
    for s in range(0, 1024, 128):
        head += det.submit(clip[s:s + 128])
    head += det.flush()
    same_keypoints(head, res.events[:1024])     # chunks of 128 = chunks of 120
    model = det.mog.export_model()
    tail = det.submit(clip[1024:1056]) + det.flush()
    model_after = det.mog.export_model()
    det.close()
    ref = D.Mog(w, h)
    ref.set_state(model, 1024)
    want = [D.blob_detect(ref.apply(f), bp)
            for f in clip[1024:1056].cpu().numpy()]
    same_keypoints(tail, want)
    same_keypoints(res.events[1024:1056], want)
    assert sum(len(k) for k in want) > 500
    assert np.array_equal(model_after.view(np.uint32),
                          ref.model().view(np.uint32))
    # how far from trivial the model is there: pixels with >= 3 components
    deep = float((model_after[:, 2, 1] > 0).mean())
    print('config C: {:.1%} of the pixels hold three or more mixture '
          'components after 1056 frames'.format(deep))
    assert deep > 0.01

    # the whole clip: the pipelined batch gives the same result, the numbers
    # are the ones bench.py reports, the arcs are the rendered flights
    again = list(sc.scan_many([clip, clip, clip]))
    sc.close()
    for other in again:
        same_keypoints(other.events, res.events)
        assert np.array_equal(other.arc_params, res.arc_params)
        assert np.array_equal(other.final_arc, res.final_arc)
    events = sum(len(k) for k in res.events)
    found, total, g = flights_recovered(spec, res, tol_px=5.0)
    print('config C: {} events, {} arcs, {}/{} flights recovered'.format(
        events, len(res.arc_params), found, total))
    assert events == CONFIG_C_EVENTS
    assert len(res.arc_params) == CONFIG_C_ARCS
    assert total > 100 and found >= 0.95 * total
    assert abs(res.g_px_per_frame_sq / g - 1.0) < 0.02


# ---------------------------------------------------------------------------
# D: 3840x2160, 240 fps, 9 balls (14 400 frames; time-segmented over GPUs)
# ---------------------------------------------------------------------------

def test_config_D_4k_step2_matches_oracle_across_a_chunk_border():
    """24 frames of the 4K clip in chunks of 16: keypoints of every frame and
    the whole 40-float model of all 8.3 M pixels, bit for bit."""
    spec = synth.CLIPS['D_4k240_9ball_long']
    w, h = spec['width'], spec['height']
    frames = synth.ClipRenderer(spec).render(3000, 24)
    got, det = detect(frames, w, h, chunk=16)
    assert det.use_tiles
    model = det.mog.export_model()
    det.close()
    want, want_model = oracle_step2(frames.cpu().numpy(), w, h)
    same_keypoints(got, want)
    assert sum(len(k) for k in want) >= 24
    assert np.array_equal(model.view(np.uint32), want_model.view(np.uint32))


def test_config_D_4k_time_segments_agree_with_the_sequential_run():
    """The layout north_star names for this configuration, two segments
    emulated on one GPU at 4K: a segment started cold 400 frames early
    against the sequential run -- masks >= 99.9 % of the pixels, events
    within 0.5 px (the level measured on MI355X is pinned below)."""
    from hawkeye_amd import dist as hkdist
    from hawkeye_amd.detect import BackgroundSubtractorMOG, EventDetector
    from hawkeye_amd.frames import SyntheticSource
    spec = dict(synth.CLIPS['D_4k240_9ball_long'], frames=1200)
    w, h = spec['width'], spec['height']
    (f0, k0, e0), (f1, k1, e1) = hkdist.segment_bounds(spec['frames'], 2, 400)
    assert (f0, k0, e0, f1, k1, e1) == (0, 0, 600, 200, 600, 1200)

    # masks: sequential model vs the model started cold at frame f1
    renderer = synth.ClipRenderer(spec)
    seq, seg = BackgroundSubtractorMOG(w, h), BackgroundSubtractorMOG(w, h)
    same = total = 0
    fg_seq = fg_both = 0
    for s in range(0, spec['frames'], 100):
        frames = renderer.render(s, 100)
        a = seq.apply_batch(frames)
        if s >= f1:
            b = seg.apply_batch(frames)
            if s >= k1:
                same += int((a == b).sum().item())
                total += a.numel()
                fg_seq += int((a == 255).sum().item())
                fg_both += int(((a == 255) & (b == 255)).sum().item())
        del frames
    seq.close()
    seg.close()
    agree = same / total
    print('4K mask agreement after a 400-frame warm-up: {:.6f}; foreground '
          'recall {:.4f}'.format(agree, fg_both / max(1, fg_seq)))
    assert agree >= 0.999                   # north_star's mask criterion

    # events: whole clip vs the two segments
    det = EventDetector(w, h)
    whole = hkdist.detect_segment(SyntheticSource(spec), det, 0, 0,
                                  spec['frames'])
    det.close()
    parts = []
    for first, keep, end in ((f0, k0, e0), (f1, k1, e1)):
        det = EventDetector(w, h)
        parts += hkdist.detect_segment(SyntheticSource(spec), det, first,
                                       keep, end)
        det.close()
    assert len(parts) == len(whole) == spec['frames']
    same_keypoints(parts[:k1], whole[:k1])      # segment 0 = sequential run
    matched = total = 0
    for a, b in zip(parts[k1:], whole[k1:]):
        total += len(b)
        if len(a) and len(b):
            d = np.hypot(b[:, None, 0] - a[None, :, 0],
                         b[:, None, 1] - a[None, :, 1])
            matched += int((d.min(axis=1) <= 0.5).sum())
    print('4K segment-2 events within 0.5 px of the sequential run: '
          '{}/{} = {:.4f}'.format(matched, total, matched / max(1, total)))
    assert total > 3000
    assert matched >= EVENT_AGREEMENT_4K * total


# measured on MI355X (this test's own print), minus a small margin
EVENT_AGREEMENT_4K = 0.995           # measured 0.9984 (masks 0.999783)


# ---------------------------------------------------------------------------
# E: 64 clips 1920x1080 @ 60 fps, 600 frames, 3..9 balls
# ---------------------------------------------------------------------------

def test_config_E_64_clips(monkeypatch):
    from hawkeye_amd.batch import ClipScanner
    specs = [synth.clip_E(i) for i in range(64)]
    w, h = specs[0]['width'], specs[0]['height']

    # clip by clip: step 2 through the 1-bit tile path and the byte-mask
    # path; three clips also against the oracle on their first frames
    per_clip = []
    for i, spec in enumerate(specs):
        frames = synth.ClipRenderer(spec).render(0, spec['frames'])
        tiles, det = detect(frames, w, h, 128, True, monkeypatch)
        assert det.use_tiles
        det.close()
        if i % 8 == 0:
            bytes_, det = detect(frames, w, h, 128, False, monkeypatch)
            assert not det.use_tiles
            det.close()
            same_keypoints(tiles, bytes_)
        if i in (0, 27, 63):
            want, _ = oracle_step2(frames[:20].cpu().numpy(), w, h)
            same_keypoints(tiles[:20], want)
        per_clip.append(tiles)
        del frames
    monkeypatch.setenv('HK_DETECT_TILES', '1')

    # the pipelined batch over all 64 clips, rendered right before they are
    # scanned: per clip the same events, in total their sum
    def lazily():
        for spec in specs:
            yield synth.ClipRenderer(spec).render(0, spec['frames'])
    sc = ClipScanner(w, h, 60.0)
    results = list(sc.scan_many(lazily(), count=64))
    assert len(results) == 64
    total = 0
    for i, (res, want) in enumerate(zip(results, per_clip)):
        same_keypoints(res.events, want)
        total += sum(len(k) for k in res.events)
    assert total == sum(len(k) for c in per_clip for k in c) > 100000

    # steps 3-4 of the batch = steps 3-4 clip by clip, and the arcs are the
    # rendered flights
    found = flights = 0
    for i in (0, 13, 38, 63):
        alone = sc.scan(synth.ClipRenderer(specs[i]).render(
            0, specs[i]['frames']))
        assert np.array_equal(alone.arc_params, results[i].arc_params)
        assert np.array_equal(alone.final_arc, results[i].final_arc)
        a, b, _ = flights_recovered(specs[i], results[i], tol_px=5.0)
        found += a
        flights += b
    sc.close()
    print('config E: {} events in 64 clips; {}/{} flights of 4 clips '
          'recovered'.format(total, found, flights))
    assert flights > 100 and found >= 0.95 * flights
