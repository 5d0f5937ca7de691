"""
Randomised step-2 parity (a short run of tests/fuzz_detect.py): random clip
sizes (odd ones included), noise levels, distractors and chunk lengths; every
frame's keypoints and the final MOG model bit for bit against the C oracle.
"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _tool():
    path = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'tests', 'fuzz_detect.py')
    spec = importlib.util.spec_from_file_location('fuzz_detect', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('seed', [5, 6])
def test_random_clips_match_oracle(seed):
    fz = _tool()
    rng = np.random.RandomState(seed)
    keypoints = 0
    for _ in range(20):
        spec = fz.random_spec(rng, 60000)
        chunk = int(rng.choice([1, 2, 7, 32, 128]))
        err, n = fz.run_case(spec, chunk)
        assert err is None, '{} (chunk {}, {})'.format(err, chunk, spec)
        keypoints += n
    assert keypoints > 0
