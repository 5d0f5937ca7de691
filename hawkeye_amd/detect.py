"""
Host-side mirror of the two OpenCV objects the reference's step 2 uses
(reference tracker.py:357-396), over device-resident frames.

    fgbg = cv2.bgsegm.createBackgroundSubtractorMOG()  ->  BackgroundSubtractorMOG
    fgmask = fgbg.apply(frame)                         ->  .apply(frame)

Frames and masks are PyTorch-ROCm uint8 tensors (plumbing only: device memory
and the current HIP stream); all arithmetic happens in libhawkeye_hip.so.
"""
import collections
import os

import numpy as np
import torch

from . import _native as nat


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class BackgroundSubtractorMOG:
    """KaewTraKulPong-Bowden mixture model with OpenCV's default parameters
    (history 200, 5 mixtures, backgroundRatio 0.7, noiseSigma 15), i.e. what
    ``cv2.bgsegm.createBackgroundSubtractorMOG()`` builds."""

    def __init__(self, width, height, device=0):
        self.width, self.height = int(width), int(height)
        self.device = torch.device('cuda', device)
        h = nat.C.c_void_p()
        nat.check(nat.lib.hk_mog_create(self.width, self.height, device,
                                        nat.C.byref(h)))
        self._h = h

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h and nat is not None and getattr(nat, 'lib', None) is not None:
            nat.lib.hk_mog_destroy(h)

    __del__ = close

    @property
    def nframes(self):
        return nat.lib.hk_mog_nframes(self._h)

    def reset(self):
        nat.check(nat.lib.hk_mog_reset(self._h, _stream_ptr(self.device)))

    def _as_device_frames(self, frames):
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        if frames.dtype != torch.uint8:
            raise TypeError('frames must be uint8 BGR')
        if frames.device != self.device:
            frames = frames.to(self.device)
        return frames

    def apply(self, frame, out=None):
        """One BGR frame [H][W][3] -> foreground mask [H][W] of {0, 255}."""
        frame = self._as_device_frames(frame)
        if tuple(frame.shape) != (self.height, self.width, 3):
            raise ValueError('frame shape {} != ({}, {}, 3)'.format(
                tuple(frame.shape), self.height, self.width))
        return self.apply_batch(frame.unsqueeze(0),
                                None if out is None else out.unsqueeze(0))[0]

    def apply_batch(self, frames, out=None):
        """Frames [T][H][W][3] -> masks [T][H][W]; identical to T apply()
        calls, in one launch with the model held in registers."""
        frames = self._as_device_frames(frames)
        if frames.dim() != 4 or tuple(frames.shape[1:]) != (
                self.height, self.width, 3):
            raise ValueError('frames shape {} != (T, {}, {}, 3)'.format(
                tuple(frames.shape), self.height, self.width))
        if frames.stride(3) != 1 or frames.stride(2) != 3:
            frames = frames.contiguous()
        t = frames.shape[0]
        if out is None:
            out = torch.empty((t, self.height, self.width), dtype=torch.uint8,
                              device=self.device)
        elif (out.dtype != torch.uint8 or out.device != self.device or
              tuple(out.shape) != (t, self.height, self.width) or
              out.stride(2) != 1):
            raise ValueError('bad `out` tensor')
        nat.check(nat.lib.hk_mog_apply_batch(
            self._h, frames.data_ptr(), frames.stride(0), frames.stride(1),
            out.data_ptr(), out.stride(0), out.stride(1), t,
            _stream_ptr(self.device)))
        return out

    @property
    def tiles_supported(self):
        """The 1-bit tile output needs frames made of 32x4 pixel tiles."""
        return bool(nat.lib.hk_mog_tiles_supported(self._h))

    def apply_batch_tiles(self, frames, tiles_ptr):
        """apply_batch() with the masks written as 1-bit tiles into the
        device buffer at `tiles_ptr` (SimpleBlobDetector.tiles_buffer)."""
        frames = self._as_device_frames(frames)
        if frames.dim() != 4 or tuple(frames.shape[1:]) != (
                self.height, self.width, 3):
            raise ValueError('frames shape {} != (T, {}, {}, 3)'.format(
                tuple(frames.shape), self.height, self.width))
        if not frames.is_contiguous():
            frames = frames.contiguous()
        nat.check(nat.lib.hk_mog_apply_batch_tiles(
            self._h, frames.data_ptr(), frames.stride(0), frames.stride(1),
            tiles_ptr, frames.shape[0], _stream_ptr(self.device)))

    def export_model(self):
        """Mixture model as float32 [H*W][5][8] in OpenCV's MixData order
        (sortKey, weight, mean[3], var[3]); for parity tests."""
        out = np.empty((self.height * self.width, 5, 8), np.float32)
        nat.check(nat.lib.hk_mog_export_model(self._h, out.ctypes.data,
                                              _stream_ptr(self.device)))
        return out


def hawkeye_blob_params(scan_height):
    """The SimpleBlobDetector parameter block the reference builds
    (reference tracker.py:359-374)."""
    p = nat.BlobParams()
    nat.check(nat.lib.hk_blob_params_hawkeye(nat.C.byref(p), int(scan_height)))
    return p


class SimpleBlobDetector:
    """``cv2.SimpleBlobDetector_create(params)`` for binary device masks.

    ``detect(masks)`` takes uint8 masks [T][H][W] (or one [H][W]) on the
    device and returns, per frame, a float32 array [n][3] of keypoints
    (x, y, size) in OpenCV's order.
    """

    def __init__(self, width, height, params=None, max_frames=64, device=0):
        self.width, self.height = int(width), int(height)
        self.max_frames = int(max_frames)
        self.device = torch.device('cuda', device)
        self.params = params if params is not None else \
            hawkeye_blob_params(height)
        h = nat.C.c_void_p()
        nat.check(nat.lib.hk_blob_create(self.width, self.height,
                                         self.max_frames,
                                         nat.C.byref(self.params), device,
                                         nat.C.byref(h)))
        self._h = h
        self._cap = self.max_frames * 128
        self.last_batch = None
        self._counts = np.empty(self.max_frames, np.int32)
        self._kp = np.empty((self._cap, 3), np.float32)
        self._centers = np.empty((self._cap, 3), np.float64)
        self._pending = 0

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h and nat is not None and getattr(nat, 'lib', None) is not None:
            nat.lib.hk_blob_destroy(h)

    __del__ = close

    def _check_masks(self, masks):
        if masks.dim() == 2:
            masks = masks.unsqueeze(0)
        if (masks.dtype != torch.uint8 or masks.device != self.device or
                masks.dim() != 3 or
                tuple(masks.shape[1:]) != (self.height, self.width)):
            raise ValueError('masks must be uint8 [T][{}][{}] on {}'.format(
                self.height, self.width, self.device))
        if masks.stride(2) != 1:
            masks = masks.contiguous()
        return masks

    def enqueue(self, masks, stream=None):
        """Start detection on up to max_frames masks (on `stream`, default
        the current one); returns immediately.  The caller must keep `masks`
        alive until finish()."""
        masks = self._check_masks(masks)
        sp = _stream_ptr(self.device) if stream is None else stream.cuda_stream
        nat.check(nat.lib.hk_blob_enqueue(
            self._h, masks.data_ptr(), masks.stride(0), masks.stride(1),
            masks.shape[0], sp))
        self._pending = masks.shape[0]
        self._keep = masks

    def tiles_buffer(self):
        """Device address of the handle's 1-bit tile buffer (room for
        max_frames frames), which BackgroundSubtractorMOG.apply_batch_tiles
        fills; enqueue_tiles() then detects on it."""
        ptr, words = nat.C.c_void_p(), nat.C.c_long()
        nat.check(nat.lib.hk_blob_tiles_buffer(self._h, nat.C.byref(ptr),
                                               nat.C.byref(words)))
        return ptr.value

    def enqueue_tiles(self, nframes, stream=None):
        sp = _stream_ptr(self.device) if stream is None else stream.cuda_stream
        nat.check(nat.lib.hk_blob_enqueue_tiles(self._h, int(nframes), sp))
        self._pending = int(nframes)
        self._keep = None

    def _split(self, flat, total):
        # one copy of the batch's rows; the per-frame arrays are views of it
        rows = flat[:total].copy()
        counts = self._counts[:self._pending].copy()
        ends = np.cumsum(counts).tolist()
        assert (ends[-1] if ends else 0) == total
        out, k = [], 0
        for e in ends:
            out.append(rows[k:e])
            k = e
        self._pending = 0
        self._keep = None
        # the same batch as two arrays, for callers that want all frames at
        # once (EventDetector.batches)
        self.last_batch = (counts, rows)
        return out

    def _finish_into(self, fn, name, dtype):
        """Calls hk_blob_finish*(); a batch with more results than the output
        array holds is kept by the library and fetched again into a larger
        one (OpenCV has no limit on the number of keypoints)."""
        total = nat.C.c_int()
        for _ in range(2):
            buf = getattr(self, name)
            status = fn(self._h, self._counts.ctypes.data, buf.ctypes.data,
                        len(buf), nat.C.byref(total))
            if status != nat.HK_ERR_CAPACITY or total.value <= len(buf):
                break
            setattr(self, name, np.empty((2 * total.value, 3), dtype))
        nat.check(status)
        return self._split(getattr(self, name), total.value)

    def finish(self):
        return self._finish_into(nat.lib.hk_blob_finish, '_kp', np.float32)

    def finish_centers(self):
        """Per-contour stage only: float64 [n][3] (x, y, radius) per frame,
        what one findBlobs() pass yields.  For tests."""
        return self._finish_into(nat.lib.hk_blob_finish_centers, '_centers',
                                 np.float64)

    def detect(self, masks):
        masks = self._check_masks(masks)
        out = []
        for s in range(0, masks.shape[0], self.max_frames):
            self.enqueue(masks[s:s + self.max_frames])
            out.extend(self.finish())
        return out


class EventDetector:
    """Step-2 arithmetic for a stream of frame chunks: MOG -> blob detector.

    ``submit(frames)`` enqueues MOG and blob detection for one chunk and
    returns the per-frame keypoints of an *earlier* chunk (three work sets
    rotate, so the blob kernels and the host-side grouping of chunks i-1 and
    i-2 overlap the MOG of chunk i); ``flush()`` returns what is still in
    flight.  Results come back in frame order: a list with one float32
    [n][3] array (x, y, size) per frame.

    MOG runs on the caller's current stream; the blob kernels of a chunk run
    on a second stream behind an event, so the (latency-bound) border tracing
    of chunk i shares the GPU with the (arithmetic-bound) MOG of chunk i+1.
    """

    def __init__(self, width, height, blob_params=None, chunk=128, device=0):
        self.width, self.height, self.chunk = int(width), int(height), \
            int(chunk)
        self.device = torch.device('cuda', device)
        self.mog = BackgroundSubtractorMOG(width, height, device)
        # DEPTH work sets rotate: MOG of chunk i is launched while the blob
        # kernels of chunks i-1 and i-2 may still be running or waiting for
        # their host-side finish
        self.blob = [SimpleBlobDetector(width, height, blob_params,
                                        max_frames=self.chunk, device=device)
                     for _ in range(self.DEPTH)]
        # The MOG kernel hands the masks over as 1-bit tiles where the frame
        # geometry allows (1/8 of the bytes for the blob scan, none written
        # as bytes); otherwise as byte images.  Same events either way.
        self.use_tiles = self.mog.tiles_supported and \
            os.environ.get('HK_DETECT_TILES', '1') != '0'
        if self.use_tiles:
            self._tiles = [b.tiles_buffer() for b in self.blob]
            self.masks = None
        else:
            self.masks = [torch.empty((self.chunk, self.height, self.width),
                                      dtype=torch.uint8, device=self.device)
                          for _ in range(self.DEPTH)]
        self._next = 0              # chunks submitted so far
        self._collected = 0         # chunks whose keypoints were returned
        # set to a list to get every collected chunk there as (tag given to
        # submit(), per-frame keypoints, (counts, rows)) instead of as
        # submit()'s / flush()'s return value: for callers that keep chunks
        # of several clips in flight (ClipScanner.detect_many)
        self.collected = None
        self._tags = collections.deque()
        # set to a list to collect (start, end, frames) event triples of the
        # MOG launches, recorded on the stream they run on
        # benchmarking: a list to which (start event, end event, frames) of
        # every mog_timing_every-th MOG launch is appended.  Two timing events
        # around a launch cost the pipeline about 0.1 ms (the launch no
        # longer overlaps its neighbours in the queue), so a timed region
        # samples launches instead of bracketing all of them
        self.mog_timing = None
        self.mog_timing_every = 1
        # set to a list to also get every collected chunk as (counts int32
        # [frames], rows float32 [keypoints][3]), in order
        self.batches = None
        self._side = torch.cuda.Stream(device=self.device)
        self._mog_done = [torch.cuda.Event() for _ in range(self.DEPTH)]

    DEPTH = int(os.environ.get('HK_DETECT_DEPTH', '3'))

    def submit(self, frames, tag=None):
        t = frames.shape[0]
        if t == 0 or t > self.chunk:
            raise ValueError('chunk of {} frames (max {})'.format(t,
                                                                   self.chunk))
        out = []
        if self._next - self._collected >= self.DEPTH:      # set still busy
            out = self._collect()
        p = self._next % self.DEPTH
        main = torch.cuda.current_stream(self.device)
        if self.use_tiles:
            def run_mog():
                self.mog.apply_batch_tiles(frames, self._tiles[p])
        else:
            masks = self.masks[p][:t]

            def run_mog():
                self.mog.apply_batch(frames, out=masks)
        if self.mog_timing is None or self._next % self.mog_timing_every:
            run_mog()
        else:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(main)
            run_mog()
            e1.record(main)
            self.mog_timing.append((e0, e1, t))
        self._mog_done[p].record(main)
        self._side.wait_event(self._mog_done[p])
        if self.use_tiles:
            self.blob[p].enqueue_tiles(t, stream=self._side)
        else:
            self.blob[p].enqueue(masks, stream=self._side)
        self._next += 1
        self._tags.append(tag)
        # keep DEPTH - 1 chunks in flight behind the one just launched
        while self._next - self._collected > self.DEPTH - 1:
            out += self._collect()
        return out

    def _collect(self):
        if self._collected >= self._next:
            return []
        p = self._collected % self.DEPTH
        self._collected += 1
        tag = self._tags.popleft()
        out = self.blob[p].finish()
        if self.batches is not None:
            self.batches.append(self.blob[p].last_batch)
        if self.collected is not None:
            self.collected.append((tag, out, self.blob[p].last_batch))
            return []
        return out

    def flush(self):
        out = []
        while self._collected < self._next:
            out += self._collect()
        return out

    def close(self):
        self.mog.close()
        for b in self.blob:
            b.close()
