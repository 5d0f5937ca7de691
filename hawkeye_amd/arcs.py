"""
Array-level interface to steps 3-4 (``hk_clip`` in include/hk.h).

``Clip`` takes the step-2 events as SoA NumPy arrays -- the same content as
``notes['meas']`` in (frame, position-in-frame) order -- and runs
build_initial_arcs (host C++) and EM_optimize (host control + HIP kernels)
natively.  ``hawkeye_amd.tracker.VideoScanner`` converts between this and the
reference's ``notes`` / ``Balltag`` / ``Ballarc`` objects.
"""
import numpy as np

from . import _native as nat

C = nat.C


def scan_params(frame_height, scan_height, fps, overrides=None):
    """hk_scan_params from reference default_scanner_params() plus the keys
    step 2 derives; ``overrides`` is a notes['scanner_params']-style dict."""
    p = nat.ScanParams()
    nat.check(nat.lib.hk_scan_params_default(C.byref(p), int(frame_height),
                                             int(scan_height), float(fps)))
    if overrides:
        for name, _ in nat.ScanParams._fields_:
            if name in overrides:
                setattr(p, name, overrides[name])
    return p


class Clip:
    def __init__(self, width, height, fps, frame_count, frame, x, y, radius,
                 total_weight=None, params=None, device=0, stream=None):
        self.frame = np.ascontiguousarray(frame, np.int32)
        n = self.frame.shape[0]
        self.x = np.ascontiguousarray(x, np.float64)
        self.y = np.ascontiguousarray(y, np.float64)
        self.radius = np.ascontiguousarray(radius, np.float64)
        self.total_weight = (np.ones(n, np.float64) if total_weight is None
                             else np.ascontiguousarray(total_weight,
                                                       np.float64))
        self.n = n
        self.params = params if params is not None else \
            scan_params(height, height, fps)
        self._stream = stream
        h = C.c_void_p()
        nat.check(nat.lib.hk_clip_create(
            int(width), int(height), float(fps), int(frame_count),
            C.byref(self.params), n, self.frame.ctypes.data,
            self.x.ctypes.data, self.y.ctypes.data, self.radius.ctypes.data,
            self.total_weight.ctypes.data, int(device), C.byref(h)))
        self._h = h
        self._cb = None

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h and nat is not None and getattr(nat, 'lib', None) is not None:
            nat.lib.hk_clip_destroy(h)

    __del__ = close

    def _stream_ptr(self):
        if self._stream is not None:
            return self._stream
        import torch
        return torch.cuda.current_stream().cuda_stream

    # -- progress / abort --------------------------------------------------

    def set_progress(self, fn):
        """fn() is called from inside the native loops; an exception raised by
        it aborts the step and is re-raised by the step method."""
        self._error = None
        if fn is None:
            self._cb = None
            nat.check(nat.lib.hk_clip_set_progress(
                self._h, nat.C.cast(None, nat.PROGRESS_FN), None))
            return

        def trampoline(_user):
            try:
                fn()
                return 0
            except BaseException as exc:       # noqa: B902 - re-raised later
                self._error = exc
                return 1
        self._cb = nat.PROGRESS_FN(trampoline)
        nat.check(nat.lib.hk_clip_set_progress(self._h, self._cb, None))

    def _run(self, status):
        if status == -5 and getattr(self, '_error', None) is not None:
            exc, self._error = self._error, None
            raise exc
        return nat.check(status)

    # -- state ---------------------------------------------------------------

    def set_state(self, camera_tilt=0.0, g=None):
        nat.check(nat.lib.hk_clip_set_state(
            self._h, float(camera_tilt), 0 if g is None else 1,
            0.0 if g is None else float(g)))

    def state(self):
        tilt, g, cm = C.c_double(), C.c_double(), C.c_double()
        has_g = C.c_int()
        nat.check(nat.lib.hk_clip_get_state(self._h, C.byref(tilt),
                                            C.byref(has_g), C.byref(g),
                                            C.byref(cm)))
        return {'camera_tilt': tilt.value,
                'g_px_per_frame_sq': g.value if has_g.value else None,
                'cm_per_pixel': cm.value if has_g.value else None}

    def set_arcs(self, params5, tag_lists, ids=None):
        a = len(tag_lists)
        params5 = np.ascontiguousarray(params5, np.float64).reshape(a, 5)
        off = np.zeros(a + 1, np.int32)
        off[1:] = np.cumsum([len(t) for t in tag_lists])
        idx = (np.concatenate([np.asarray(t, np.int32) for t in tag_lists])
               if a and off[-1] else np.zeros(0, np.int32))
        idx = np.ascontiguousarray(idx, np.int32)
        ids_arr = (np.ascontiguousarray(ids, np.int32) if ids is not None
                   else None)
        nat.check(nat.lib.hk_clip_set_arcs(
            self._h, a, params5.ctypes.data,
            None if ids_arr is None else ids_arr.ctypes.data,
            off.ctypes.data, idx.ctypes.data))

    def arcs(self):
        """(params [A][5] = a, b, c, e, f_peak; ids [A]; list of member tag
        index arrays)."""
        a = nat.lib.hk_clip_arc_count(self._h)
        total = nat.lib.hk_clip_arc_tag_total(self._h)
        params = np.empty((a, 5), np.float64)
        ids = np.empty(a, np.int32)
        off = np.empty(a + 1, np.int32)
        idx = np.empty(max(total, 1), np.int32)
        nat.check(nat.lib.hk_clip_get_arcs(self._h, params.ctypes.data,
                                           ids.ctypes.data, off.ctypes.data,
                                           idx.ctypes.data))
        return params, ids, [idx[off[k]:off[k + 1]].copy() for k in range(a)]

    def tag_state(self):
        alive = np.empty(self.n, np.uint8)
        final = np.empty(self.n, np.int32)
        nat.check(nat.lib.hk_clip_get_tag_state(self._h, alive.ctypes.data,
                                                final.ctypes.data))
        return alive.astype(bool), final

    # -- steps ---------------------------------------------------------------

    def build_initial_arcs(self):
        self._run(nat.lib.hk_clip_build_initial_arcs(self._h))

    def em_optimize(self):
        self._run(nat.lib.hk_clip_em_optimize(self._h, self._stream_ptr()))

    @staticmethod
    def em_optimize_many(clips, stream=None):
        """Step 4 of several clips of one device in lockstep
        (hk_clips_em_optimize): every EM kernel is launched once over the
        arcs of all of them.  Each clip ends up as em_optimize() leaves it."""
        clips = list(clips)
        if not clips:
            return
        handles = (C.c_void_p * len(clips))(*[c._h for c in clips])
        if stream is None:
            stream = clips[0]._stream_ptr()
        status = nat.lib.hk_clips_em_optimize(handles, len(clips), stream)
        if status == -5:
            for c in clips:
                if getattr(c, '_error', None) is not None:
                    exc, c._error = c._error, None
                    raise exc
        nat.check(status)

    def calculate_weights(self):
        self._run(nat.lib.hk_clip_calculate_weights(self._h,
                                                    self._stream_ptr()))

    def fit_arcs(self):
        self._run(nat.lib.hk_clip_fit_arcs(self._h, self._stream_ptr()))

    def estimate_camera_tilt(self):
        self._run(nat.lib.hk_clip_estimate_camera_tilt(self._h,
                                                       self._stream_ptr()))

    def weights(self):
        """E-step weights as (tag [m], arc index [m], weight [m])."""
        n = nat.check(nat.lib.hk_clip_get_weights(self._h, None, None, None,
                                                  0, self._stream_ptr()))
        tag = np.empty(max(n, 1), np.int32)
        arc = np.empty(max(n, 1), np.int32)
        w = np.empty(max(n, 1), np.float64)
        nat.check(nat.lib.hk_clip_get_weights(
            self._h, tag.ctypes.data, arc.ctypes.data, w.ctypes.data, n,
            self._stream_ptr()))
        return tag[:n], arc[:n], w[:n]

    def eval_arc(self, index, requirepeak=False, checkaccel=True):
        return nat.check(nat.lib.hk_clip_eval_arc(self._h, int(index),
                                                  int(requirepeak),
                                                  int(checkaccel)))

    def member_distances(self):
        total = nat.lib.hk_clip_arc_tag_total(self._h)
        d = np.empty(max(total, 1), np.float64)
        nat.check(nat.lib.hk_clip_member_distances(self._h, d.ctypes.data,
                                                   self._stream_ptr()))
        return d[:total]
