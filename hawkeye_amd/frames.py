"""
Frame sources for step 2.

Video decode stays on the host and is out of scope (north_star); step 2 only
needs "the next frames as BGR uint8 on the device".  A source exposes

    fps, width, height, frame_count_estimate
    read(max_frames) -> uint8 tensor [T][H][W][3] on the device, T == 0 at end
    seek(frame)      -> optional; needed only for time-segmented multi-GPU runs

``open_source(path)`` resolves, in order: a source registered for that path
(``register_source``; how tests, benchmarks and host decoders hand frames in),
then ``cv2.VideoCapture`` when OpenCV is importable (the reference's own
decode path, reference tracker.py:262-271, 315-330, 383), else the same
"Error opening video file" failure the reference raises.
"""
import os

import numpy as np
import torch

_REGISTRY = {}


def register_source(path, factory):
    """Associate ``path`` (as given to VideoScanner) with a zero-argument
    callable returning a fresh source."""
    _REGISTRY[os.path.abspath(path)] = factory


def unregister_source(path):
    _REGISTRY.pop(os.path.abspath(path), None)


class SourceError(Exception):
    pass


class ArraySource:
    """Frames held in host (NumPy) or device (torch) memory."""

    def __init__(self, frames, fps, device=0):
        self.device = torch.device('cuda', device)
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        if frames.dtype != torch.uint8 or frames.dim() != 4 or \
                frames.shape[3] != 3:
            raise ValueError('frames must be uint8 [T][H][W][3] (BGR)')
        self._frames = frames
        self.fps = float(fps)
        self.frame_count_estimate = int(frames.shape[0])
        self.height, self.width = int(frames.shape[1]), int(frames.shape[2])
        self._pos = 0

    def read(self, max_frames):
        chunk = self._frames[self._pos:self._pos + max_frames]
        self._pos += chunk.shape[0]
        if chunk.device != self.device:
            chunk = chunk.to(self.device, non_blocking=True)
        return chunk

    def seek(self, frame):
        self._pos = int(frame)

    def close(self):
        self._frames = None


class PinnedHostSource:
    """Frames in page-locked host memory, streamed to the device ahead of use.

    This is the ingest path for a host decoder (reference tracker.py:383
    decodes with cv2.VideoCapture on the CPU): the decoder writes BGR frames
    into a pinned buffer, and read() hands out device chunks while the next
    chunk is already crossing PCIe on a copy stream.  Three device buffers
    rotate, so a chunk is never overwritten while kernels enqueued on the
    caller's stream may still read it.

    ``frames``: uint8 tensor [T][H][W][3], pinned (``torch.Tensor.pin_memory``
    or ``PinnedHostSource.alloc``); a pageable tensor / NumPy array is pinned
    once up front (one extra host copy).
    """

    BUFFERS = 3

    @staticmethod
    def alloc(nframes, height, width):
        """Page-locked uint8 [T][H][W][3] for a decoder to fill."""
        return torch.empty((nframes, height, width, 3), dtype=torch.uint8,
                           pin_memory=True)

    def __init__(self, frames, fps, device=0, chunk=128):
        self.device = torch.device('cuda', device)
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        if frames.dtype != torch.uint8 or frames.dim() != 4 or \
                frames.shape[3] != 3 or frames.is_cuda:
            raise ValueError('frames must be host uint8 [T][H][W][3] (BGR)')
        if not frames.is_pinned():
            frames = frames.contiguous().pin_memory()
        self._frames = frames
        self.fps = float(fps)
        self.frame_count_estimate = int(frames.shape[0])
        self.height, self.width = int(frames.shape[1]), int(frames.shape[2])
        self._chunk = int(chunk)
        self._copy = torch.cuda.Stream(device=self.device)
        self._buf = [torch.empty((self._chunk, self.height, self.width, 3),
                                 dtype=torch.uint8, device=self.device)
                     for _ in range(self.BUFFERS)]
        self._ready = [torch.cuda.Event() for _ in range(self.BUFFERS)]
        self._pos = 0           # next frame to hand out
        self._k = 0             # index of the next chunk handed out
        self._staged = None     # (chunk index, first frame, count) in flight

    def _stage(self, k, first, reuse_after):
        """Start copying frames [first, first+n) into buffer k % BUFFERS."""
        n = min(self._chunk, self.frame_count_estimate - first)
        if n <= 0:
            self._staged = None
            return
        b = k % self.BUFFERS
        self._copy.wait_event(reuse_after)
        with torch.cuda.stream(self._copy):
            self._buf[b][:n].copy_(self._frames[first:first + n],
                                   non_blocking=True)
            self._ready[b].record(self._copy)
        self._staged = (k, first, n)

    def read(self, max_frames):
        if max_frames != self._chunk:
            raise ValueError('PinnedHostSource was built for chunks of {} '
                             'frames'.format(self._chunk))
        cur = torch.cuda.current_stream(self.device)
        # everything enqueued so far on the caller's stream (the kernels that
        # read earlier chunks) must finish before a buffer is written again
        fence = torch.cuda.Event()
        fence.record(cur)
        if self._staged is None or self._staged[1] != self._pos:
            self._stage(self._k, self._pos, fence)      # first read / seek
        if self._staged is None:
            return torch.empty((0, self.height, self.width, 3),
                               dtype=torch.uint8, device=self.device)
        k, first, n = self._staged
        b = k % self.BUFFERS
        self._pos = first + n
        self._k = k + 1
        self._stage(self._k, self._pos, fence)          # prefetch the next
        cur.wait_event(self._ready[b])
        return self._buf[b][:n]

    def seek(self, frame):
        self._pos = int(frame)

    def close(self):
        torch.cuda.current_stream(self.device).synchronize()
        self._copy.synchronize()
        self._frames = None
        self._buf = []


class SyntheticSource:
    """A clip of hawkeye_amd.synth rendered on the device chunk by chunk."""

    def __init__(self, spec, device=0, first=0, count=None):
        from . import synth
        self._renderer = synth.ClipRenderer(spec, device=device)
        self.fps = float(spec['fps'])
        self.width, self.height = spec['width'], spec['height']
        self._first = first
        self._end = spec['frames'] if count is None else first + count
        self.frame_count_estimate = self._end - first
        self._pos = first

    def read(self, max_frames):
        n = min(max_frames, self._end - self._pos)
        if n <= 0:
            return torch.empty((0, self.height, self.width, 3),
                               dtype=torch.uint8,
                               device=self._renderer.device)
        out = self._renderer.render(self._pos, n)
        self._pos += n
        return out

    def seek(self, frame):
        self._pos = self._first + int(frame)

    def close(self):
        self._renderer = None


class OpenCVSource:
    """cv2.VideoCapture decode on the host, uploaded in chunks."""

    def __init__(self, path, device=0):
        import cv2
        self._cv2 = cv2
        self.device = torch.device('cuda', device)
        self._cap = cv2.VideoCapture(path)
        if not self._cap.isOpened():
            raise SourceError(path)
        self.fps = self._cap.get(cv2.CAP_PROP_FPS)
        self.frame_count_estimate = int(self._cap.get(
            cv2.CAP_PROP_FRAME_COUNT))
        self.width = int(self._cap.get(cv2.CAP_PROP_FRAME_WIDTH))
        self.height = int(self._cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
        self._reads = 0
        self._ended = False
        # failed reads are retried until more reads than this were made.  The
        # reference compares against notes['frame_count_estimate'] -- the
        # count of the SOURCE video from step 1, also when a scan video is
        # read (tracker.py:307, 390): VideoScanner sets it; on its own a
        # source uses its file's estimate
        self.read_limit = self.frame_count_estimate

    def _decode(self, max_frames):
        """Up to max_frames decoded frames (host arrays): the retry rule of
        reference tracker.py:382-392."""
        got = []
        while (len(got) < max_frames and not self._ended and
               self._cap.isOpened()):
            ok, frame = self._cap.read()
            self._reads += 1
            if not ok:
                if self._reads > self.read_limit:
                    self._ended = True      # the reference's loop ends here
                continue
            got.append(frame)
        return got

    def read(self, max_frames):
        got = self._decode(max_frames)
        if not got:
            return torch.empty((0, self.height, self.width, 3),
                               dtype=torch.uint8, device=self.device)
        host = torch.from_numpy(np.stack(got)).pin_memory()
        return host.to(self.device, non_blocking=True)

    def seek(self, frame):
        self._cap.set(self._cv2.CAP_PROP_POS_FRAMES, int(frame))
        self._reads = int(frame)
        self._ended = False

    def close(self):
        self._cap.release()


def open_source(path, device=0):
    factory = _REGISTRY.get(os.path.abspath(path))
    if factory is not None:
        return factory()
    try:
        import cv2  # noqa: F401
    except ImportError:
        raise SourceError(path)
    return OpenCVSource(path, device=device)
