"""
CPU: host logic of the frame sources and of the step-5 stand-in, no GPU.
OpenCVSource is driven with a stub cv2 module (cv2 is not in the image): the
retry rule for failed reads must follow reference tracker.py:382-392 with the
limit VideoScanner hands over (notes['frame_count_estimate'], tracker.py:307).
"""
import sys
import types
import warnings

import numpy as np
import pytest


class FakeCapture:
    """Yields `pattern`: True = a frame, False = a failed read; then fails
    forever."""

    def __init__(self, pattern, count_prop, w=8, h=6):
        self.pattern, self.pos = list(pattern), 0
        self.props = {1: 30.0, 2: float(count_prop), 3: float(w), 4: float(h)}
        self.w, self.h = w, h

    def isOpened(self):
        return True

    def get(self, prop):
        return self.props[prop]

    def read(self):
        ok = self.pos < len(self.pattern) and self.pattern[self.pos]
        self.pos += 1
        if not ok:
            return False, None
        return True, np.full((self.h, self.w, 3), self.pos, np.uint8)

    def release(self):
        pass


@pytest.fixture
def stub_cv2(monkeypatch):
    mod = types.ModuleType('cv2')
    mod.CAP_PROP_FPS, mod.CAP_PROP_FRAME_COUNT = 1, 2
    mod.CAP_PROP_FRAME_WIDTH, mod.CAP_PROP_FRAME_HEIGHT = 3, 4
    mod.CAP_PROP_POS_FRAMES = 5
    mod.captures = []

    def video_capture(path):
        cap = FakeCapture(mod.pattern, mod.count)
        mod.captures.append(cap)
        return cap
    mod.VideoCapture = video_capture
    monkeypatch.setitem(sys.modules, 'cv2', mod)
    return mod


def decode_all(src):
    frames = []
    while True:
        got = src._decode(4)
        if not got:
            return frames
        frames += got


def test_failed_reads_are_retried_up_to_the_estimate(stub_cv2):
    from hawkeye_amd.frames import OpenCVSource
    # two failed reads in the middle, container says 10 frames, 8 decode
    stub_cv2.pattern = [True, True, False, True, False] + [True] * 5
    stub_cv2.count = 10
    src = OpenCVSource('/nonexistent.mov', device=0)
    frames = decode_all(src)
    assert len(frames) == 8
    # reads go on until MORE reads than the estimate were made: 11
    assert stub_cv2.captures[0].pos == 11
    # the frames that decoded, in order (the stub writes the read number)
    assert [int(f[0, 0, 0]) for f in frames] == [1, 2, 4, 6, 7, 8, 9, 10]


def test_retry_limit_comes_from_the_notes_not_from_the_scan_video(stub_cv2):
    """A scan video whose container reports fewer frames than the source
    video's estimate: the reference keeps retrying against the SOURCE's
    estimate (notes['frame_count_estimate']), so a failed read late in the
    scan video does not end the scan early."""
    from hawkeye_amd.frames import OpenCVSource
    stub_cv2.pattern = [True] * 6 + [False] + [True] * 3
    stub_cv2.count = 5                     # the scan video's own (low) count
    src = OpenCVSource('/nonexistent_scan.mov', device=0)
    assert src.read_limit == 5
    own = len(decode_all(src))
    assert own == 6                        # its own estimate: stops at the gap
    src = OpenCVSource('/nonexistent_scan.mov', device=0)
    src.read_limit = 12                    # what VideoScanner.detect_objects sets
    assert len(decode_all(src)) == 9


def test_step5_without_detector_is_recorded_and_warned():
    from hawkeye_amd.tracker import VideoScanner
    sc = VideoScanner('/nonexistent/clip.mov')
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        sc.detect_juggler()
    assert sc.notes['body'] == {} and sc.notes['body_source'] == 'estimated'
    assert any('body detector' in str(w.message) for w in caught)
    sc = VideoScanner('/nonexistent/clip.mov',
                      body_detector=lambda s: {3: (1, 2, 3, 4, True)})
    sc.detect_juggler()
    assert sc.notes['body_source'] == 'detector' and 3 in sc.notes['body']
