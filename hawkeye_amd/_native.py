"""
ctypes binding of libhawkeye_hip.so (C ABI declared in include/hk.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C
hawkeye_amd/csrc``).  There is no fallback: if the shared object is missing or
does not load, importing this module raises, and every product entry point
that needs it fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HK_NATIVE_LIB: another build of the same library (instrumented builds of
# tools/); never a different implementation -- there is no CPU fallback
LIB_PATH = os.environ.get('HK_NATIVE_LIB') or \
    os.path.join(_HERE, 'libhawkeye_hip.so')


class NativeError(RuntimeError):
    """A libhawkeye_hip call returned a negative status."""

    def __init__(self, status, message):
        super().__init__('libhawkeye_hip error {}: {}'.format(status, message))
        self.status = status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            '{} not found: build it with `python -c "import __graft_entry__ '
            'as g; g.build()"` or `make -C hawkeye_amd/csrc`; there is no '
            'CPU fallback'.format(LIB_PATH))
    # PyTorch-ROCm ships its own libamdhip64.so.7; import it first so that
    # this library binds to the same HIP runtime instance and device pointers
    # of torch tensors are valid here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return C.CDLL(LIB_PATH)


lib = _load()

c_void_pp = C.POINTER(C.c_void_p)


def _sig(name, argtypes, restype=C.c_int):
    try:
        fn = getattr(lib, name)
    except AttributeError:
        raise ImportError(
            '{} does not export {}(): it is not a build of this tree\'s '
            'include/hk.h{}; rebuild it (`make -C hawkeye_amd/csrc`)'.format(
                LIB_PATH, name, ' (HK_NATIVE_LIB is set)'
                if os.environ.get('HK_NATIVE_LIB') else '')) from None
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


_sig('hk_last_error', [], C.c_char_p)
_sig('hk_version', [])
_sig('hk_device_count', [])
_sig('hk_release_cached_memory', [])
_sig('hk_set_host_threads', [C.c_int])
_sig('hk_step3_help', [C.c_int])
_sig('hk_step3_device_built', [])
_sig('hk_set_self_check', [C.c_int])

_sig('hk_mog_create', [C.c_int, C.c_int, C.c_int, c_void_pp])
_sig('hk_mog_selfcheck_math', [C.c_int, C.POINTER(C.c_longlong)])
_sig('hk_mog_destroy', [C.c_void_p])
_sig('hk_mog_reset', [C.c_void_p, C.c_void_p])
_sig('hk_mog_nframes', [C.c_void_p])
_sig('hk_mog_apply', [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                      C.c_void_p])
_sig('hk_mog_apply_batch', [C.c_void_p, C.c_void_p, C.c_long, C.c_long,
                            C.c_void_p, C.c_long, C.c_long, C.c_int,
                            C.c_void_p])
_sig('hk_mog_tiles_supported', [C.c_void_p])
_sig('hk_mog_apply_batch_tiles', [C.c_void_p, C.c_void_p, C.c_long, C.c_long,
                                  C.c_void_p, C.c_int, C.c_void_p])
_sig('hk_mog_export_model', [C.c_void_p, C.c_void_p, C.c_void_p])
_sig('hk_mog_apply_batch_static', [C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_long, C.c_long,
                                   C.c_void_p, C.c_long, C.c_long, C.c_int,
                                   C.c_void_p])


class BlobParams(C.Structure):
    """hk_blob_params (include/hk.h); field names follow
    cv2.SimpleBlobDetector_Params."""
    _fields_ = [('thresholdStep', C.c_float), ('minThreshold', C.c_float),
                ('maxThreshold', C.c_float), ('minRepeatability', C.c_int),
                ('minDistBetweenBlobs', C.c_float), ('filterByColor', C.c_int),
                ('blobColor', C.c_int), ('filterByArea', C.c_int),
                ('minArea', C.c_float), ('maxArea', C.c_float),
                ('filterByCircularity', C.c_int),
                ('minCircularity', C.c_float), ('maxCircularity', C.c_float),
                ('filterByInertia', C.c_int), ('filterByConvexity', C.c_int)]


_sig('hk_blob_params_hawkeye', [C.POINTER(BlobParams), C.c_int])
_sig('hk_blob_create', [C.c_int, C.c_int, C.c_int, C.POINTER(BlobParams),
                        C.c_int, c_void_pp])
_sig('hk_blob_destroy', [C.c_void_p])
_sig('hk_blob_enqueue', [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int,
                         C.c_void_p])
_sig('hk_blob_tiles_buffer', [C.c_void_p, C.POINTER(C.c_void_p),
                              C.POINTER(C.c_long)])
_sig('hk_blob_enqueue_tiles', [C.c_void_p, C.c_int, C.c_void_p])
_sig('hk_blob_finish', [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                        C.POINTER(C.c_int)])
_sig('hk_blob_detect', [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int,
                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                        C.POINTER(C.c_int)])
_sig('hk_blob_finish_centers', [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.POINTER(C.c_int)])

_sig('hk_render_frames', [C.c_int, C.c_int, C.c_float, C.c_uint, C.c_float,
                          C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int,
                          C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long,
                          C.c_void_p])


class ScanParams(C.Structure):
    """hk_scan_params (include/hk.h): the scanner constants of reference
    tracker.py:2510-2570 that steps 3-4 read, plus the keys step 2 derives."""
    _fields_ = [('sigmasq', C.c_double), ('max_frame_gap_in_arc', C.c_int),
                ('g_window', C.c_double), ('min_tags_to_curve_fit', C.c_int),
                ('default_frame_height_cm', C.c_double),
                ('max_frame_height_cm', C.c_double),
                ('default_fps', C.c_double), ('max_fps', C.c_double),
                ('min_tags_per_arc', C.c_int),
                ('max_distance_pixels', C.c_double),
                ('radius_window', C.c_double)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)

_sig('hk_scan_params_default', [C.POINTER(ScanParams), C.c_int, C.c_int,
                                C.c_double])
_sig('hk_clip_create', [C.c_int, C.c_int, C.c_double, C.c_int,
                        C.POINTER(ScanParams), C.c_int, C.c_void_p,
                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                        C.c_int, c_void_pp])
_sig('hk_clip_destroy', [C.c_void_p])
_sig('hk_clip_set_progress', [C.c_void_p, PROGRESS_FN, C.c_void_p])
_sig('hk_clip_set_state', [C.c_void_p, C.c_double, C.c_int, C.c_double])
_sig('hk_clip_get_state', [C.c_void_p, C.POINTER(C.c_double),
                           C.POINTER(C.c_int), C.POINTER(C.c_double),
                           C.POINTER(C.c_double)])
_sig('hk_clip_set_arcs', [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                          C.c_void_p, C.c_void_p])
_sig('hk_clip_arc_count', [C.c_void_p])
_sig('hk_clip_arc_tag_total', [C.c_void_p])
_sig('hk_clip_get_arcs', [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_void_p])
_sig('hk_clip_get_tag_state', [C.c_void_p, C.c_void_p, C.c_void_p])
_sig('hk_clip_build_initial_arcs', [C.c_void_p])
_sig('hk_clip_em_optimize', [C.c_void_p, C.c_void_p])
_sig('hk_clips_em_optimize', [C.POINTER(C.c_void_p), C.c_int, C.c_void_p])
_sig('hk_clip_calculate_weights', [C.c_void_p, C.c_void_p])
_sig('hk_clip_fit_arcs', [C.c_void_p, C.c_void_p])
_sig('hk_clip_estimate_camera_tilt', [C.c_void_p, C.c_void_p])
_sig('hk_clip_get_weights', [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int, C.c_void_p])
_sig('hk_clip_eval_arc', [C.c_void_p, C.c_int, C.c_int, C.c_int])
_sig('hk_clip_member_distances', [C.c_void_p, C.c_void_p, C.c_void_p])


HK_ERR_CAPACITY = -3        # include/hk.h


def check(status):
    """Raise NativeError for a negative status, else return the status."""
    if status < 0:
        raise NativeError(status, lib.hk_last_error().decode('utf-8',
                                                             'replace'))
    return status
