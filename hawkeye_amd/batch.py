"""
Array-level driver of steps 2-4 for one clip or a batch of clips.

``VideoScanner`` (hawkeye_amd/tracker.py) is the reference's object API: it
builds ``Balltag`` / ``Ballarc`` objects and the ``notes`` dictionary.  This
module is what sits underneath when many clips are analysed (BASELINE.json
configs[4]) or when only the arrays are wanted: frames in, events and arcs
out as NumPy arrays, through the same C ABI.

    scanner = ClipScanner(width, height, fps, device=0)
    result = scanner.scan(frames)                  # one clip
    for result in scanner.scan_many(clips): ...    # pipelined batch

Step 2 is GPU work (MOG + blob kernels) with a thin host loop; steps 3-4 are
host work (multi-threaded step 3, merge / prune control flow) around small
EM kernels.  ``scan_many`` therefore runs them as a pipeline: one worker thread
detects the clips ahead on its own HIP stream, and a pool of builder threads
(one per spare host core) takes a detected clip each through step 3 and step 4
(EM kernels on a stream per thread); a builder thread without a clip helps the
step-3 builds in progress.  Results are the same as scanning the clips one by
one, and come in the same order.
"""
import collections
import os
import queue
import threading
import time

import numpy as np
import torch

from . import _native as nat
from .arcs import Clip, scan_params
from .detect import EventDetector, hawkeye_blob_params


def host_core_share():
    """Host cores this process may plan on: the cores it is allowed to run on,
    divided among the ranks of the node (one process per GPU), capped by
    OMP_NUM_THREADS (the GPU boxes export their CPU share there) and by 16.
    torch.distributed.run exports OMP_NUM_THREADS=1 to every rank when the
    variable was unset; with several local ranks a value of 1 is taken to be
    that default, not a statement about the machine."""
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))
    share = max(1, avail // ranks)
    try:
        omp = int(os.environ.get('OMP_NUM_THREADS', '0') or 0)
    except ValueError:
        omp = 0
    if omp > 1 or (omp == 1 and ranks == 1):
        share = min(share, omp)
    return max(1, min(share, 16))


class ClipResult:
    """Events and arcs of one clip.

    events         list, one float32 [n][3] array (x, y, size) per frame:
                   the keypoints of step 2 in scan-video coordinates
    frame, x, y, radius   the same events as notes['meas'] holds them, SoA
    arc_params     float64 [A][5] = a, b, c, e, f_peak (sorted by f_peak)
    arc_ids, arc_members   id_ per arc; member event indices per arc
    alive, final_arc       per event: kept by clean_notes / index of tag.arc
    camera_tilt, g_px_per_frame_sq, cm_per_pixel   notes scalars
    """
    __slots__ = ('events', 'frame', 'x', 'y', 'radius', 'frame_count',
                 'arc_params', 'arc_ids', 'arc_members', 'alive', 'final_arc',
                 'camera_tilt', 'g_px_per_frame_sq', 'cm_per_pixel',
                 'stage_seconds')


class FrameEvents(list):
    """The per-frame keypoint arrays of a clip (a plain list), with the same
    data as two arrays: `counts` int [frames] and `rows` float32
    [keypoints][3], all frames end to end."""
    counts = None
    rows = None


def events_to_soa(per_frame, frame_width, frame_height, scan_width,
                  scan_height):
    """Keypoints -> the SoA arrays of notes['meas']: the float64 expressions
    of reference tracker.py:335-342, 413-417 (the blob *diameter* goes into
    Balltag.radius, as the reference does)."""
    if getattr(per_frame, 'rows', None) is not None:
        # (a few array operations instead of 1800 small ones under the
        # interpreter lock, which the detection thread needs every chunk)
        counts = np.asarray(per_frame.counts, np.int64)
        kp = per_frame.rows.astype(np.float64)
    else:
        counts = np.fromiter((len(k) for k in per_frame), np.int64,
                             len(per_frame))
        kp = (np.concatenate(per_frame) if counts.sum() else
              np.zeros((0, 3), np.float32)).astype(np.float64)
    frame = np.repeat(np.arange(len(per_frame), dtype=np.int32), counts)
    scale = frame_height / scan_height
    crop_w = frame_height * (scan_width / scan_height)
    pad_left = (frame_width - crop_w) / 2
    return frame, pad_left + kp[:, 0] * scale, kp[:, 1] * scale, \
        kp[:, 2] * scale


class ClipScanner:
    def __init__(self, width, height, fps, device=0, chunk=360,
                 frame_size=None, params=None):
        """width x height: the scanned frames.  chunk: frames per MOG launch
        (a launch ends with a tail in which its slowest tiles run alone: 360
        frames per launch instead of 120 took the 1080p benchmark clip's
        detection from 7.75 to 6.9 ms).  frame_size = (W, H) of the
        source video if a down-scaled scan video is analysed (reference
        tracker.py:310-342), else None."""
        self.width, self.height, self.fps = int(width), int(height), \
            float(fps)
        self.frame_width, self.frame_height = (
            (self.width, self.height) if frame_size is None else
            (int(frame_size[0]), int(frame_size[1])))
        self.device = int(device)
        self.chunk = int(chunk)
        self.params = params if params is not None else scan_params(
            self.frame_height, self.height, self.fps)
        self._det = EventDetector(self.width, self.height,
                                  blob_params=hawkeye_blob_params(self.height),
                                  chunk=self.chunk, device=self.device)
        dev = torch.device('cuda', self.device)
        self._detect_stream = torch.cuda.Stream(device=dev)
        self._em_stream = torch.cuda.Stream(device=dev)
        # detection lanes: (detector, stream); lane 0 is the scanner's own,
        # scan_many's further detection threads get theirs on first use
        self._lanes = [(self._det, self._detect_stream)]
        self._lanes_lock = threading.Lock()

    def _lane(self, i):
        with self._lanes_lock:
            while len(self._lanes) <= i:
                det = EventDetector(
                    self.width, self.height,
                    blob_params=hawkeye_blob_params(self.height),
                    chunk=self.chunk, device=self.device)
                stream = torch.cuda.Stream(
                    device=torch.device('cuda', self.device))
                self._lanes.append((det, stream))
            return self._lanes[i]

    @property
    def detector(self):
        return self._det

    def close(self):
        det, self._det = getattr(self, '_det', None), None
        if det is not None:
            det.close()
        lanes, self._lanes = getattr(self, '_lanes', None) or [], []
        for other, _ in lanes[1:]:
            other.close()

    __del__ = close

    # -- step 2 ------------------------------------------------------------

    def detect(self, frames):
        """frames: uint8 [T][H][W][3] BGR on the device, or a frame source
        (hawkeye_amd.frames) with read().  Returns the per-frame keypoints."""
        det = self._det
        out = FrameEvents()
        det.batches = []
        # the caller may have produced the frames on its own stream a moment
        # ago (rendered, decoded, copied): order the detection stream after it
        self._detect_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._detect_stream):
            det.mog.reset()
            chunks = self._chunks_of(frames)
            try:
                for chunk in chunks:
                    out.extend(det.submit(chunk))
                out.extend(det.flush())
                batches = det.batches
            finally:
                det.batches = None
        if batches:
            out.counts = np.concatenate([c for c, _ in batches])
            out.rows = np.concatenate([r for _, r in batches])
        else:
            out.counts = np.zeros(0, np.int32)
            out.rows = np.zeros((0, 3), np.float32)
        assert len(out.counts) == len(out)
        return out

    def detect_many(self, clips, emit, lane=0):
        """Step 2 of one clip after the other without draining the chunk
        pipeline in between: the first MOG launches of a clip are queued
        while the blob kernels and the host-side finish of the previous
        clip's last chunks are still to come (a drained pipeline costs about
        1 ms of GPU idle time per clip of the 1080p benchmark, 11 %).  Calls
        emit(per_frame, seconds) once per clip, in order; `seconds` is the
        time the clip added to the stream of results.  Same events as
        detect() clip by clip.  The next clip is taken from `clips` before
        the previous one's last chunks are collected: an iterator that takes
        long to produce a clip delays those results by that time.
        `lane`: which detector and stream to use (0 = the scanner's own; see
        scan_many)."""
        det, detect_stream = self._lane(lane)
        det.collected = []
        open_clips = collections.deque()
        last_done = [time.perf_counter()]

        def drain():
            for state, out, batch in det.collected:
                state['out'].extend(out)
                state['batches'].append(batch)
                state['collected'] += 1
            del det.collected[:]
            while open_clips and open_clips[0]['closed'] and \
                    open_clips[0]['collected'] == open_clips[0]['submitted']:
                state = open_clips.popleft()
                out = state['out']
                if state['batches']:
                    out.counts = np.concatenate(
                        [c for c, _ in state['batches']])
                    out.rows = np.concatenate([r for _, r in state['batches']])
                else:
                    out.counts = np.zeros(0, np.int32)
                    out.rows = np.zeros((0, 3), np.float32)
                assert len(out.counts) == len(out)
                now = time.perf_counter()
                since = max(last_done[0], state['t0'])
                last_done[0] = now
                emit(out, now - since)

        try:
            for frames in clips:
                state = {'out': FrameEvents(), 'batches': [], 'submitted': 0,
                         'collected': 0, 'closed': False,
                         't0': time.perf_counter()}
                open_clips.append(state)
                detect_stream.wait_stream(
                    torch.cuda.current_stream(self.device))
                with torch.cuda.stream(detect_stream):
                    det.mog.reset()
                    chunks = self._chunks_of(frames)
                    for chunk in chunks:
                        det.submit(chunk, tag=state)
                        state['submitted'] += 1
                        if det.collected:
                            drain()
                state['closed'] = True
                drain()
            with torch.cuda.stream(detect_stream):
                det.flush()
            drain()
        except BaseException:
            # leave the detector empty, and let the clips that were complete
            # but for their last chunks reach the caller before the error
            try:
                # the clip that failed, if the error came from one: a clip
                # whose chunks were all submitted is complete (the error then
                # came from the caller's iterator or from a later clip) and
                # is drained like the ones before it
                if open_clips and not open_clips[-1]['closed']:
                    open_clips.pop()
                with torch.cuda.stream(detect_stream):
                    det.flush()
                drain()
            except Exception:       # noqa: BLE001 - the first error counts
                pass
            raise
        finally:
            det.collected = None

    def _chunks_of(self, frames):
        """A clip as chunks of at most self.chunk frames.  A tensor is cut
        into equal parts (1800 frames: 15 x 120, not 14 x 128 + 8): a short
        last launch gives the GPU less to do than the host needs for a
        chunk, and the queue runs dry at every clip's end."""
        if hasattr(frames, 'read'):
            return self._from_source(frames)
        total = int(frames.shape[0])
        parts = max(1, -(-total // self.chunk))
        size = max(1, -(-total // parts))
        # whole groups of 4 frames where the chunk length allows: what is
        # left over of a launch goes through the frame-at-a-time kernel,
        # which reads and writes the whole model for every frame
        if ((size + 3) & ~3) <= self.chunk:
            size = (size + 3) & ~3
        return (frames[s:s + size] for s in range(0, total, size))

    def _from_source(self, src):
        while True:
            chunk = src.read(self.chunk)
            if chunk.shape[0] == 0:
                return
            yield chunk

    # -- steps 3-4 ---------------------------------------------------------

    def build_arcs(self, per_frame, result=None, em_stream=None):
        """Step 3 on the keypoints of one clip.  Returns (result, clip); the
        open native clip goes to refine_arcs(), whose kernels run on
        `em_stream` (default: this scanner's EM stream)."""
        if em_stream is None:
            em_stream = self._em_stream
        res = result if result is not None else ClipResult()
        t0 = time.perf_counter()
        c0 = time.thread_time()
        res.events = per_frame
        res.frame_count = len(per_frame)
        res.frame, res.x, res.y, res.radius = events_to_soa(
            per_frame, self.frame_width, self.frame_height, self.width,
            self.height)
        clip = Clip(self.frame_width, self.frame_height, self.fps,
                    res.frame_count, res.frame, res.x, res.y, res.radius,
                    params=self.params, device=self.device,
                    stream=em_stream.cuda_stream)
        try:
            clip.build_initial_arcs()
        except BaseException:
            clip.close()
            raise
        self._stage(res, 'step3', time.perf_counter() - t0)
        # CPU time of the calling thread alone (helpers and step-3 worker
        # threads not included): what a one-thread build costs its core
        self._stage(res, 'step3_cpu', time.thread_time() - c0)
        return res, clip

    def refine_arcs(self, res, clip):
        """Step 4 and the read-back of its results; closes the clip."""
        t0 = time.perf_counter()
        c0 = time.thread_time()
        try:
            clip.em_optimize()
            res.arc_params, res.arc_ids, res.arc_members = clip.arcs()
            res.alive, res.final_arc = clip.tag_state()
            st = clip.state()
        finally:
            clip.close()
        res.camera_tilt = st['camera_tilt']
        res.g_px_per_frame_sq = st['g_px_per_frame_sq']
        res.cm_per_pixel = st['cm_per_pixel']
        self._stage(res, 'step4', time.perf_counter() - t0)
        self._stage(res, 'step4_cpu', time.thread_time() - c0)
        return res

    def refine_arcs_many(self, pairs, em_stream=None):
        """Step 4 of several clips together: `pairs` = [(result, clip)] as
        build_arcs() returns them (all clips on this scanner's device).  The
        EM kernels run once per phase over the arcs of all clips
        (hk_clips_em_optimize); every result is what refine_arcs() gives for
        the clip on its own.  Closes the clips."""
        if em_stream is None:
            em_stream = self._em_stream
        t0 = time.perf_counter()
        c0 = time.thread_time()
        clips = [clip for _, clip in pairs]
        try:
            Clip.em_optimize_many(clips, stream=em_stream.cuda_stream)
            for res, clip in pairs:
                res.arc_params, res.arc_ids, res.arc_members = clip.arcs()
                res.alive, res.final_arc = clip.tag_state()
                st = clip.state()
                res.camera_tilt = st['camera_tilt']
                res.g_px_per_frame_sq = st['g_px_per_frame_sq']
                res.cm_per_pixel = st['cm_per_pixel']
        finally:
            for clip in clips:
                clip.close()
        # a clip's share of the call
        wall = (time.perf_counter() - t0) / len(pairs)
        cpu = (time.thread_time() - c0) / len(pairs)
        for res, _ in pairs:
            self._stage(res, 'step4', wall)
            self._stage(res, 'step4_cpu', cpu)
            self._stage(res, 'step4_batch', len(pairs))
        return [res for res, _ in pairs]

    @staticmethod
    def _stage(res, name, seconds):
        stages = getattr(res, 'stage_seconds', None) or {}
        stages[name] = seconds
        res.stage_seconds = stages

    # -- whole clips -------------------------------------------------------

    def scan(self, frames):
        t0 = time.perf_counter()
        per_frame = self.detect(frames)
        res = ClipResult()
        self._stage(res, 'detect', time.perf_counter() - t0)
        return self.refine_arcs(*self.build_arcs(per_frame, res))

    @staticmethod
    def plan_host_stage(count=None, cores=None):
        """(clips in step 3 at a time, worker threads per clip) for a batch of
        `count` clips (None = long or unknown) on `cores` host cores (None =
        this rank's share), one of which is left to the detection loop.

        A step-3 build runs seeds speculatively ahead of its committer, and
        speculation re-does work: measured on MI355X + 16 cores (1080p60
        9-ball clip, tools/sweep_builders.sh) one clip costs 147 ms of one
        core, but 183 / 227 / 260 ms of core time on 2 / 7 / 16 threads.  So
        every clip is built on one thread of its own, as many clips at a time
        as there are cores (or clips), and threads that have no clip -- while
        the batch fills up, and when it drains -- help the builds in progress
        (hk_step3_help) instead of owning a share of a build."""
        if cores is None:
            cores = host_core_share()
        # one core for the detection loop (and one for the step-4 thread where
        # HK_EM_THREAD=1 asks for it: it sleeps through its stream waits, but
        # its host work is ~5 ms per clip)
        spare = 2 if os.environ.get('HK_EM_THREAD', '0') != '0' else 1
        return max(1, cores - spare), 1

    def scan_many(self, clips, host_threads=None, builders=None, count=None,
                  lookahead=None):
        """Generator over the results of `clips` (an iterable of frame
        tensors / sources of this scanner's size), in order.  Detection of
        later clips (GPU, one worker thread) runs beside steps 3-4 of earlier
        ones (`builders` threads, a clip each; step 3 on `host_threads` worker
        threads per clip).  With lookahead, detection can run on
        `HK_DETECT_LANES` threads, each with a detector and a stream of its
        own, taking clips in turn: a MOG launch ends with a tail of slow
        waves that leaves part of the chip idle, and the other lane's launch
        fills it (tools/probe/two_detectors.py, detection only: 7.7 ms per
        clip with one lane, 6.2-7.2 with two).  The default is one lane: with
        16 host cores per GPU the builders bound a batch, not detection, and
        a second lane gains nothing (60-clip batch 164-171 k frames/s with
        one lane, 164-165 k with two); with more cores per GPU set 2.  `count` = number of clips if `clips` cannot tell
        (a generator); unused today, kept for callers that pass it.
        `lookahead` (default on): detection starts on the next clip before
        the last chunks of the previous one are collected (detect_many)."""
        if lookahead is None:
            lookahead = os.environ.get('HK_DETECT_LOOKAHEAD', '1') != '0'
        if count is None and hasattr(clips, '__len__'):
            count = len(clips)
        if builders is None and host_threads is None:
            builders, host_threads = self.plan_host_stage(count)
        elif builders is None:
            builders = max(1, (host_core_share() - 1) // int(host_threads))
        builders = max(1, int(builders))
        lanes = 1
        if lookahead:
            lanes = max(1, int(os.environ.get('HK_DETECT_LANES', '1')))
            if count is not None:
                lanes = max(1, min(lanes, int(count)))
        detected = queue.Queue(maxsize=builders)
        built = queue.Queue(maxsize=builders)
        # Step 4.  Default: every builder thread takes its clip on through
        # step 4 (the host half of step 4, ~5 ms per clip, then runs on as
        # many cores as there are builders), but all of them on ONE stream,
        # the scanner's EM stream: with a stream per builder, 15 + 3 streams
        # shared the 4 hardware queues, and an EM kernel that landed in the
        # queue of the detection stream waited behind whole MOG launches --
        # step 4 took 25 ms per clip inside a 60-clip batch, 12.5 ms on the
        # shared stream (and detection 7.4 instead of 8.1 ms; MI355X, 16
        # cores; HK_EM_SHARED_STREAM=0 for the old way).
        # HK_EM_THREAD=1: the clips whose step 3 is complete go to ONE thread
        # that runs step 4 of all that are waiting in lockstep, every EM
        # kernel launched once over the arcs of all of them
        # (hk_clips_em_optimize).  Same results; measured slower on 16 cores
        # per GPU (the thread costs a builder, and the last clips of a batch
        # queue up behind it: 148-154 k against 163-172 k frames/s over 20
        # clips), so it is the option, not the default.
        em_thread = os.environ.get('HK_EM_THREAD', '0') != '0'
        em_max = max(1, int(os.environ.get('HK_EM_BATCH', '8')))
        shared_em_stream = os.environ.get('HK_EM_SHARED_STREAM', '1') != '0'
        refine_q = queue.Queue()
        stop = threading.Event()
        END = object()

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def get(q):
            while not stop.is_set():
                try:
                    return q.get(timeout=0.05)
                except queue.Empty:
                    pass
            return END

        source = iter(clips)
        src_lock = threading.Lock()
        next_seq = [0]
        live = [lanes]
        failed = [False]

        def detect_stage(lane):
            mine = collections.deque()  # this lane's clips not yet delivered
            try:
                torch.cuda.set_device(self.device)

                def until_stopped():
                    while True:
                        with src_lock:
                            if stop.is_set() or failed[0]:
                                return
                            try:
                                frames = next(source)
                            except StopIteration:
                                return
                            mine.append(next_seq[0])
                            next_seq[0] += 1
                        yield frames

                def emit(per_frame, seconds):
                    res = ClipResult()
                    self._stage(res, 'detect', seconds)
                    if put(detected, (mine[0], per_frame, res)):
                        mine.popleft()

                if lookahead:
                    self.detect_many(until_stopped(), emit, lane)
                else:
                    for frames in until_stopped():
                        t0 = time.perf_counter()
                        per_frame = self.detect(frames)
                        emit(per_frame, time.perf_counter() - t0)
            except BaseException as exc:    # noqa: B902 - re-raised below
                # reported at the oldest clip this lane still owes (the one
                # that failed, or one before it that is lost with it); the
                # other lanes take no further clips
                with src_lock:
                    failed[0] = True
                    if mine:
                        at = mine[0]
                    else:
                        at = next_seq[0]
                        next_seq[0] += 1
                put(built, (at, exc))
            finally:
                with src_lock:
                    live[0] -= 1
                    last = live[0] == 0
                if last:
                    for _ in range(builders):
                        put(detected, END)

        # A builder thread with no clip of its own helps the step-3 builds
        # the other threads are running (hk_step3_help: it runs seeds ahead
        # of that build's committer; results do not depend on it), so the
        # host cores stay busy while the batch fills up and while it drains.
        HELP_SEEDS = 512
        # a builder thread without a clip helps whenever any other is
        # building (measured on MI355X + 16 cores, 1080p clip C: 20-clip batch
        # 91.5 k -> 105 k frames/s, 60-clip batch 128 k -> 131 k against
        # helping only while fewer than half of the builders are busy)
        HELP_BUSY_MAX = float(os.environ.get('HK_HELP_BUSY_MAX', '1.0'))
        building = [0]
        lock = threading.Lock()

        def next_clip():
            """The next detected clip; helps running builds while waiting."""
            while not stop.is_set():
                try:
                    return detected.get_nowait()
                except queue.Empty:
                    pass
                if building[0] >= HELP_BUSY_MAX * builders or \
                        not nat.lib.hk_step3_help(HELP_SEEDS):
                    try:
                        return detected.get(timeout=0.002)
                    except queue.Empty:
                        pass
            return END

        def build_stage():
            try:
                torch.cuda.set_device(self.device)
                # step 4 of a clip follows its step 3 on the same thread, on
                # a stream of that thread: the last clips of a batch finish
                # step 3 close together, and their step 4s then run side by
                # side instead of queueing up on one thread
                # (HK_EM_SHARED_STREAM=0: a stream per builder thread)
                em_stream = self._em_stream if shared_em_stream else \
                    torch.cuda.Stream(device=self.device)
                while True:
                    item = next_clip()
                    if item is END:
                        break
                    seq, per_frame, res = item
                    with lock:
                        building[0] += 1
                    try:
                        done = self.build_arcs(per_frame, res, em_stream)
                    except BaseException as exc:    # noqa: B902
                        put(built, (seq, exc))
                        return
                    finally:
                        with lock:
                            building[0] -= 1
                    if em_thread:
                        refine_q.put((seq,) + done)
                        continue
                    try:
                        done = self.refine_arcs(*done)
                    except BaseException as exc:    # noqa: B902
                        put(built, (seq, exc))
                        return
                    if not put(built, (seq, done)):
                        return
                # every clip has been taken by some builder (END follows the
                # last one): help those still building
                while not stop.is_set():
                    with lock:
                        if building[0] == 0:
                            break
                    if not nat.lib.hk_step3_help(HELP_SEEDS):
                        time.sleep(0.0005)
            finally:
                if em_thread:
                    refine_q.put(END)
                else:
                    put(built, END)

        def em_stage():
            """Takes the clips that have finished step 3 -- all that are
            waiting, up to em_max -- through step 4 together."""
            ended = 0
            held = []           # native clips not yet closed
            try:
                torch.cuda.set_device(self.device)
                em_stream = torch.cuda.Stream(device=self.device)
                while ended < builders:
                    items = []
                    try:
                        items.append(refine_q.get(timeout=0.05))
                    except queue.Empty:
                        if stop.is_set():
                            break
                        continue
                    while len(items) < em_max:
                        try:
                            items.append(refine_q.get_nowait())
                        except queue.Empty:
                            break
                    ended += sum(1 for it in items if it is END)
                    items = [it for it in items if it is not END]
                    if not items:
                        continue
                    held = [clip for _, _, clip in items]
                    if stop.is_set():
                        break
                    try:
                        self.refine_arcs_many([(res, clip)
                                               for _, res, clip in items],
                                              em_stream)
                    except BaseException as exc:    # noqa: B902
                        put(built, (min(seq for seq, _, _ in items), exc))
                        return
                    held = []
                    for seq, res, _ in items:
                        if not put(built, (seq, res)):
                            return
            finally:
                for clip in held:
                    clip.close()
                while True:         # clips nobody will refine any more
                    try:
                        it = refine_q.get_nowait()
                    except queue.Empty:
                        break
                    if it is not END:
                        it[2].close()
                put(built, END)

        if host_threads is None:    # builders given: share the cores out
            host_threads = max(1, (host_core_share() - 1) // builders)
        nat.check(nat.lib.hk_set_host_threads(int(host_threads)))
        workers = [threading.Thread(target=detect_stage, args=(lane,),
                                    daemon=True) for lane in range(lanes)] + \
            [threading.Thread(target=build_stage, daemon=True)
             for _ in range(builders)]
        if em_thread:
            workers.append(threading.Thread(target=em_stage, daemon=True))
        enders = 1 if em_thread else builders
        for w in workers:
            w.start()
        waiting = {}            # seq -> built clip or exception, out of order
        try:
            want = 0
            ended = 0
            while True:
                while want in waiting:
                    item = waiting.pop(want)
                    if isinstance(item, BaseException):
                        raise item
                    want += 1
                    yield item
                if ended == enders:
                    break
                item = built.get()
                if item is END:
                    ended += 1
                else:
                    waiting[item[0]] = item[1]
            for item in waiting.values():       # after a gap: an error
                if isinstance(item, BaseException):
                    raise item
        finally:
            stop.set()
            for w in workers:
                w.join()
            nat.check(nat.lib.hk_set_host_threads(0))
