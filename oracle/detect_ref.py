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
