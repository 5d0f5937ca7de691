"""
ORACLE -- test infrastructure only (see detect_ref.c).  ctypes/NumPy wrapper
over oracle/_build/libhkref.so.  Never imported by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libhkref.so')


class BlobParams(C.Structure):
    _fields_ = [('thresholdStep', C.c_float), ('minThreshold', C.c_float),
                ('maxThreshold', C.c_float), ('minRepeatability', C.c_int),
                ('minDistBetweenBlobs', C.c_float), ('filterByColor', C.c_int),
                ('blobColor', C.c_int), ('filterByArea', C.c_int),
                ('minArea', C.c_float), ('maxArea', C.c_float),
                ('filterByCircularity', C.c_int),
                ('minCircularity', C.c_float), ('maxCircularity', C.c_float)]


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.hkref_mog_create.restype = C.c_void_p
        L.hkref_mog_create.argtypes = [C.c_int, C.c_int]
        L.hkref_mog_destroy.argtypes = [C.c_void_p]
        L.hkref_mog_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_long,
                                      C.c_void_p, C.c_long]
        L.hkref_mog_nframes.argtypes = [C.c_void_p]
        L.hkref_mog_model.restype = C.POINTER(C.c_float)
        L.hkref_mog_model.argtypes = [C.c_void_p]
        L.hkref_mog_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hkref_mog_alpha.restype = C.c_float
        L.hkref_mog_alpha.argtypes = [C.c_int]
        L.hkref_find_contours.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_long,
            C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_long)),
            C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_int)]
        L.hkref_free.argtypes = [C.c_void_p]
        L.hkref_blob_params_hawkeye.argtypes = [C.POINTER(BlobParams), C.c_int]
        L.hkref_find_blobs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long,
                                       C.POINTER(BlobParams), C.c_double,
                                       C.c_void_p, C.c_int]
        L.hkref_blob_detect.argtypes = [C.c_void_p, C.c_int, C.c_int,
                                        C.c_long, C.POINTER(BlobParams),
                                        C.c_void_p, C.c_int]
        _lib = L
    return _lib


class Mog:
    """cv2.bgsegm.createBackgroundSubtractorMOG() with default arguments."""

    K = 5

    def __init__(self, width, height):
        self.w, self.h = width, height
        self._h = lib().hkref_mog_create(width, height)
        if not self._h:
            raise MemoryError

    def apply(self, frame):
        """frame: uint8 [H][W][3] BGR -> uint8 mask [H][W] of {0, 255}."""
        frame = np.ascontiguousarray(frame, np.uint8)
        assert frame.shape == (self.h, self.w, 3)
        mask = np.empty((self.h, self.w), np.uint8)
        lib().hkref_mog_apply(self._h, frame.ctypes.data, self.w * 3,
                              mask.ctypes.data, self.w)
        return mask

    def model(self):
        """Copy of the mixture model, float32 [H*W][K][8] =
        (sortKey, weight, mean[3], var[3])."""
        p = lib().hkref_mog_model(self._h)
        n = self.h * self.w * self.K * 8
        return np.ctypeslib.as_array(p, (n,)).reshape(
            self.h * self.w, self.K, 8).copy()

    def set_state(self, model, nframes):
        """Continue from `model` (as model() returns it) after `nframes`
        frames."""
        model = np.ascontiguousarray(model, np.float32)
        assert model.size == self.h * self.w * self.K * 8
        lib().hkref_mog_set_state(self._h, model.ctypes.data, int(nframes))

    def __del__(self):
        if getattr(self, '_h', None):
            lib().hkref_mog_destroy(self._h)
            self._h = None


def hawkeye_params(scan_height):
    p = BlobParams()
    lib().hkref_blob_params_hawkeye(C.byref(p), scan_height)
    return p


def find_contours(img):
    """List of (is_hole, int32 [n][2] points) in cv2.findContours(RETR_LIST,
    CHAIN_APPROX_NONE) order."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    pts, start, hole = (C.POINTER(C.c_int)(), C.POINTER(C.c_long)(),
                        C.POINTER(C.c_int)())
    n = C.c_int()
    rc = lib().hkref_find_contours(img.ctypes.data, w, h, w, C.byref(pts),
                                   C.byref(start), C.byref(hole), C.byref(n))
    if rc != 0:
        raise MemoryError
    out = []
    for i in range(n.value):
        a, b = start[i], start[i + 1]
        arr = np.ctypeslib.as_array(pts, (2 * b,))[2 * a:2 * b]
        out.append((bool(hole[i]), arr.reshape(-1, 2).copy()))
    for p in (pts, start, hole):
        lib().hkref_free(p)
    return out


def find_blobs(img, params, thresh=50.0, cap=65536):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((cap, 3), np.float64)
    n = lib().hkref_find_blobs(img.ctypes.data, w, h, w, C.byref(params),
                               thresh, out.ctypes.data, cap)
    if n < 0:
        raise MemoryError
    assert n <= cap
    return out[:n].copy()


def blob_detect(img, params, cap=65536):
    """float32 [n][3] keypoints (x, y, size) in cv2 order."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((cap, 3), np.float32)
    n = lib().hkref_blob_detect(img.ctypes.data, w, h, w, C.byref(params),
                                out.ctypes.data, cap)
    if n < 0:
        raise MemoryError
    assert n <= cap
    return out[:n].copy()


def detect_clip(frames, params, threads=1, block=64):
    """Step-2 events of a clip held on the host (uint8 [T][H][W][3]): MOG then
    the blob detector, frame by frame.  With threads > 1 the work is spread the
    way OpenCV spreads it: the MOG update over bands of rows (MOGInvoker under
    parallel_for_, bgfg_gaussmix.cpp; pixels are independent, so the masks are
    the single-thread ones bit for bit), and the per-frame blob detection over
    frames.  ctypes drops the GIL inside the C calls, so plain Python threads
    do.  Returns a list of float32 [n][3] arrays."""
    frames = np.ascontiguousarray(frames, np.uint8)
    T, H, W = frames.shape[:3]
    if threads <= 1:
        mog = Mog(W, H)
        return [blob_detect(mog.apply(f), params) for f in frames]
    from concurrent.futures import ThreadPoolExecutor
    bands = min(threads, H)
    edges = [(H * b) // bands for b in range(bands + 1)]
    mogs = [Mog(W, edges[b + 1] - edges[b]) for b in range(bands)]
    masks = np.empty((min(block, T), H, W), np.uint8)
    L = lib()

    def mog_band(b, first, count):
        r0, r1 = edges[b], edges[b + 1]
        for t in range(count):
            L.hkref_mog_apply(mogs[b]._h, frames[first + t, r0:r1].ctypes.data,
                              W * 3, masks[t, r0:r1].ctypes.data, W)

    out = []
    with ThreadPoolExecutor(threads) as pool:
        for first in range(0, T, block):
            count = min(block, T - first)
            list(pool.map(lambda b: mog_band(b, first, count), range(bands)))
            out.extend(pool.map(lambda t: blob_detect(masks[t], params),
                                range(count)))
    return out
