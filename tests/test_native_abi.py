"""CPU-only: the C-ABI library loads and exports every entry point that
include/hk.h declares (no compute calls; there is no GPU here)."""
import ctypes
import os
import re

from hawkeye_amd import _native as nat

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(REPO, 'include', 'hk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(hk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert 'hk_mog_apply' in names and len(names) >= 8
    lib = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_error_channel_without_gpu():
    assert nat.lib.hk_version() >= 1
    # a bad argument is rejected before any HIP call
    h = ctypes.c_void_p()
    rc = nat.lib.hk_mog_create(0, 10, 0, ctypes.byref(h))
    assert rc == -1
    assert b'bad size' in nat.lib.hk_last_error()
