"""
``VideoScanner`` with the reference's Python surface (reference
hawkeye/tracker.py:20-242, 2510-2616) and the hot path -- step 2 detection,
step 3 preliminary arcs, step 4 EM -- running in libhawkeye_hip.so.

What is kept from the reference, because callers depend on it
(reference worker.py:718-778, track.py:56-66, widgets.py:128-550):
  * constructor and ``process()`` signatures, step numbering, the per-step
    ``<base>_notes{N}.pkl`` files, ``CURRENT_NOTES_VERSION``
  * the ``notes`` dictionary schema (reference tracker.py:35-112) and every
    key steps 1-4 write, including the scanner_params keys derived in step 2
  * ``callback(step, maxsteps)`` once per frame in step 2 and ``callback()``
    during steps 3-4, invoked on the calling thread; an exception raised by
    the callback propagates out of ``process()`` (that is how the GUI aborts)
  * progress text through ``print`` gated by ``verbosity``
  * ``ScannerException`` for an unopenable video

Step 5 (YOLO body detection) stays host code outside this build: a detector
can be plugged in (``body_detector``), else body positions are estimated in
step 6 the way the reference does for frames its detector missed.  Step 6
(juggling analysis, hawkeye_amd/juggling.py) is the reference's host-side
post-processing of the final arcs, restated and checked against golden
vectors from the reference.
"""
import os
import pickle
from math import degrees

import numpy as np

from .types import Balltag, Ballarc


class ScannerException(Exception):
    def __init__(self, message=None):
        super().__init__(message)


class VideoScanner:

    CURRENT_NOTES_VERSION = 4

    # frames handed to the device per MOG/blob launch in step 2
    DETECT_CHUNK = 128

    def __init__(self, filename, scanvideo=None, params=None, notes=None,
                 device=0, body_detector=None):
        self.device = device
        self.body_detector = body_detector
        if notes is None:
            self.notes = {
                'version': VideoScanner.CURRENT_NOTES_VERSION,
                'source': os.path.abspath(filename),
                'scanvideo': (os.path.abspath(scanvideo)
                              if scanvideo is not None else None),
                'scanner_params': (VideoScanner.default_scanner_params()
                                   if params is None else params),
                'step': 0,
            }
        else:
            self.notes = notes

    # ------------------------------------------------------------------
    #  orchestration (reference tracker.py:153-242)
    # ------------------------------------------------------------------

    def process(self, steps=(1, 6), readnotes=False, writenotes=False,
                notesdir=None, callback=None, verbosity=0):
        self._callback = callback
        self._verbosity = verbosity
        if verbosity >= 1:
            print('Video scanner starting...')

        if readnotes or writenotes:
            srcdir, srcname = os.path.split(self.notes['source'])
            stem = os.path.splitext(srcname)[0]
            if notesdir is None:
                where = srcdir
            elif os.path.isabs(notesdir):
                where = notesdir
            else:
                where = os.path.join(srcdir, notesdir)

        first, last = steps
        if first in (2, 3, 4, 5, 6):
            if readnotes:
                self.notes = VideoScanner.read_notes(os.path.join(
                    where, '{}_notes{}.pkl'.format(stem, first - 1)))
        else:
            first = 1

        runners = {1: self.get_video_metadata,
                   2: lambda: self.detect_objects(display=False),
                   3: self.build_initial_arcs,
                   4: self.EM_optimize,
                   5: lambda: self.detect_juggler(display=False),
                   6: self.analyze_juggling}
        for step in range(first, last + 1):
            if step in runners:
                runners[step]()
            self.notes['step'] = step

        if writenotes:
            name = ('{}_notes{}.pkl'.format(stem, last)
                    if last in (1, 2, 3, 4, 5) else '{}_notes.pkl'.format(stem))
            VideoScanner.write_notes(self.notes, os.path.join(where, name))
        if verbosity >= 1:
            print('Video scanner done')

    def _progress(self, *args):
        cb = getattr(self, '_callback', None)
        if cb is not None:
            cb(*args)

    def _say(self, level, text):
        if getattr(self, '_verbosity', 0) >= level:
            print(text)

    # ------------------------------------------------------------------
    #  step 1 (reference tracker.py:248-281); decode stays on the host
    # ------------------------------------------------------------------

    def _open(self, path):
        from . import frames
        try:
            return frames.open_source(path, device=self.device)
        except frames.SourceError:
            raise ScannerException('Error opening video file {}'.format(path))

    def get_video_metadata(self):
        notes = self.notes
        self._say(1, 'Getting metadata for video {}'.format(notes['source']))
        src = self._open(notes['source'])
        notes['fps'] = src.fps
        notes['frame_width'] = src.width
        notes['frame_height'] = src.height
        notes['frame_count_estimate'] = src.frame_count_estimate
        src.close()
        self._say(2, 'width = {}, height = {}, fps = {}'.format(
            notes['frame_width'], notes['frame_height'], notes['fps']))
        self._say(2, 'estimated frame count = {}\n'.format(
            notes['frame_count_estimate']))

    # ------------------------------------------------------------------
    #  step 2 (reference tracker.py:287-444)
    # ------------------------------------------------------------------

    def detect_objects(self, display=False):
        from .detect import EventDetector, hawkeye_blob_params
        notes = self.notes
        notes['meas'] = dict()
        self._say(1, 'Object detection starting...')

        fps = notes['fps']
        frame_w, frame_h = notes['frame_width'], notes['frame_height']
        frame_est = notes['frame_count_estimate']
        scanvideo = notes['scanvideo']
        if scanvideo is None:
            self._say(2, 'Scanning from video {}'.format(notes['source']))
            src = self._open(notes['source'])
            scan_w, scan_h = frame_w, frame_h
        else:
            self._say(2, 'Scanning from video {}'.format(scanvideo))
            src = self._open(scanvideo)
            scan_w, scan_h = src.width, src.height
            self._say(2, 'width = {}, height = {}'.format(scan_w, scan_h))
        if hasattr(src, 'read_limit'):
            src.read_limit = frame_est      # reference tracker.py:307, 390
        if (src.width, src.height) != (scan_w, scan_h):
            raise ScannerException(
                'video is {}x{} but notes say {}x{}'.format(
                    src.width, src.height, scan_w, scan_h))

        # the same Python-float expressions as reference tracker.py:335-342
        scale = frame_h / scan_h
        crop_w = frame_h * (scan_w / scan_h)
        pad_left = (frame_w - crop_w) / 2

        sp = notes['scanner_params']
        notes['camera_tilt'] = 0.0
        sp['min_tags_per_arc'] = (sp['min_tags_per_arc_high_fps']
                                  if fps >= 29
                                  else sp['min_tags_per_arc_low_fps'])
        sp['max_distance_pixels'] = (sp['max_distance_pixels_480'] *
                                     frame_h / 480)
        sp['radius_window'] = (sp['radius_window_high_res']
                               if scan_h >= 480
                               else sp['radius_window_low_res'])

        bp = hawkeye_blob_params(scan_h)
        hi = scan_h >= 480
        bp.minArea = sp['min_blob_area_high_res' if hi
                        else 'min_blob_area_low_res']
        bp.maxArea = sp['max_blob_area_high_res' if hi
                        else 'max_blob_area_low_res']
        det = EventDetector(scan_w, scan_h, blob_params=bp,
                            chunk=self.DETECT_CHUNK, device=self.device)

        framenum = 0
        tag_count = 0

        def absorb(per_frame):
            nonlocal framenum, tag_count
            for kps in per_frame:
                tags = []
                for kx, ky, ksize in kps.tolist():
                    tags.append(Balltag(framenum, pad_left + kx * scale,
                                        ky * scale, ksize * scale, 1.0))
                notes['meas'][framenum] = tags
                tag_count += len(tags)
                framenum += 1
                self._progress(framenum, frame_est)

        try:
            while True:
                chunk = src.read(self.DETECT_CHUNK)
                if chunk.shape[0] == 0:
                    break
                absorb(det.submit(chunk))
            absorb(det.flush())
        finally:
            det.close()
            src.close()

        notes['frame_count'] = framenum
        self._say(2, 'actual frame count = {}'.format(framenum))
        self._say(1, 'Object detection done: {} detections\n'.format(
            tag_count))

    # ------------------------------------------------------------------
    #  notes <-> native clip
    # ------------------------------------------------------------------

    def _clip_from_notes(self):
        """SoA copy of notes['meas'] in (frame, list position) order, the
        native clip built on it and the flat list of Balltag objects."""
        from . import arcs as native
        notes = self.notes
        fcount = notes['frame_count']
        flat = []
        for f in range(fcount):
            flat.extend(notes['meas'][f])
        n = len(flat)
        frame = np.fromiter((t.frame for t in flat), np.int32, n)
        x = np.fromiter((t.x for t in flat), np.float64, n)
        y = np.fromiter((t.y for t in flat), np.float64, n)
        r = np.fromiter((t.radius for t in flat), np.float64, n)
        tw = np.fromiter((t.total_weight for t in flat), np.float64, n)
        sp = notes['scanner_params']
        params = native.scan_params(notes['frame_height'],
                                    notes['frame_height'], notes['fps'],
                                    overrides=sp)
        clip = native.Clip(notes['frame_width'], notes['frame_height'],
                           notes['fps'], fcount, frame, x, y, r, tw,
                           params=params, device=self.device)
        clip.set_state(notes.get('camera_tilt', 0.0),
                       notes.get('g_px_per_frame_sq'))
        if getattr(self, '_callback', None) is not None:
            clip.set_progress(lambda: self._callback())
        return clip, flat

    def _push_arcs(self, clip, flat, arcs):
        index = {id(t): i for i, t in enumerate(flat)}
        params = [[a.a, a.b, a.c, a.e, a.f_peak] for a in arcs]
        lists = [sorted(index[id(t)] for t in a.tags) for a in arcs]
        ids = [a.id_ if a.id_ is not None else k + 1
               for k, a in enumerate(arcs)]
        clip.set_arcs(np.array(params, np.float64).reshape(len(arcs), 5),
                      lists, ids)

    # ------------------------------------------------------------------
    #  step 3 (reference tracker.py:450-485)
    # ------------------------------------------------------------------

    def build_initial_arcs(self):
        notes = self.notes
        self._say(1, 'Build initial arcs starting...')
        clip, flat = self._clip_from_notes()
        clip.build_initial_arcs()
        state = clip.state()
        if state['g_px_per_frame_sq'] is not None:
            notes['g_px_per_frame_sq'] = state['g_px_per_frame_sq']
            notes['cm_per_pixel'] = state['cm_per_pixel']
            self._say(2, 'g (pixels/frame^2) = {:.6f}, cm/pixel = {:.6f}'
                      .format(notes['g_px_per_frame_sq'],
                              notes['cm_per_pixel']))
        params, ids, members = clip.arcs()
        arcs = []
        for p, i, m in zip(params, ids, members):
            arc = Ballarc()
            arc.a, arc.b, arc.c, arc.e, arc.f_peak = (float(v) for v in p)
            arc.id_ = int(i)
            arc.tags = {flat[j] for j in m}
            for t in arc.tags:
                t.weight = {arc: 1.0}
            arcs.append(arc)
        notes['arcs'] = arcs
        clip.close()
        self._say(1, 'Build initial arcs done: {} arcs created\n'.format(
            len(arcs)))

    # ------------------------------------------------------------------
    #  step 4 (reference tracker.py:912-990)
    # ------------------------------------------------------------------

    def EM_optimize(self):
        notes = self.notes
        arcs = notes['arcs']
        before = len(arcs)
        self._say(1, 'EM optimization starting...')
        clip, flat = self._clip_from_notes()
        self._push_arcs(clip, flat, arcs)
        clip.em_optimize()

        notes['camera_tilt'] = clip.state()['camera_tilt']
        self._say(2, '  camera tilt = {:.6f} degrees'.format(
            degrees(notes['camera_tilt'])))
        params, ids, members = clip.arcs()
        alive, final = clip.tag_state()
        clip.close()

        # keep the surviving Ballarc objects (and the list object) alive
        by_id = {a.id_: a for a in arcs}
        kept = []
        for p, i, m in zip(params, ids, members):
            arc = by_id[int(i)]
            arc.a, arc.b, arc.c, arc.e, arc.f_peak = (float(v) for v in p)
            arc.tags = {flat[j] for j in m}
            arc.__dict__.pop('done', None)
            kept.append(arc)
        arcs[:] = kept
        for i, tag in enumerate(flat):
            tag.__dict__.pop('weight', None)
            tag.__dict__.pop('done', None)
            if alive[i]:
                tag.arc = kept[final[i]]
        # drop removed tags from notes['meas'] (reference tracker.py:1368-1373)
        dead = {id(flat[i]) for i in range(len(flat)) if not alive[i]}
        if dead:
            for f in range(notes['frame_count']):
                notes['meas'][f] = [t for t in notes['meas'][f]
                                    if id(t) not in dead]
        self._say(1, 'EM done: {} arcs before, {} after\n'.format(
            before, len(arcs)))

    # The EM building blocks under the reference's names, on notes objects.

    def _with_arcs(self, arcs):
        clip, flat = self._clip_from_notes()
        self._push_arcs(clip, flat, arcs)
        return clip, flat

    def fit_arcs(self, arcs):
        """Least-squares fit of each arc with unit weights on arc.tags
        (reference tracker.py:698-798 for tags without E-step weights)."""
        clip, _ = self._with_arcs(arcs)
        clip.fit_arcs()
        params, _, _ = clip.arcs()
        for arc, p in zip(arcs, params):
            arc.a, arc.b, arc.c, arc.e, arc.f_peak = (float(v) for v in p)
        clip.close()

    def estimate_camera_tilt(self, arcs):
        clip, _ = self._with_arcs(arcs)
        clip.estimate_camera_tilt()
        self.notes['camera_tilt'] = clip.state()['camera_tilt']
        clip.close()

    def calculate_weights(self, arcs):
        """E-step: sets tag.weight = {arc: w} on every tag and rebuilds
        arc.tags (reference tracker.py:992-1039)."""
        clip, flat = self._with_arcs(arcs)
        clip.calculate_weights()
        tag, arc_no, w = clip.weights()
        _, _, members = clip.arcs()
        clip.close()
        for t in flat:
            t.weight = dict()
        for arc in arcs:
            arc.tags = set()
        for i, k, v in zip(tag.tolist(), arc_no.tolist(), w.tolist()):
            flat[i].weight[arcs[k]] = v
        for arc, m in zip(arcs, members):
            arc.tags = {flat[j] for j in m}

    def eval_arc(self, arc, requirepeak=False, checkaccel=True):
        clip, _ = self._with_arcs([arc])
        code = clip.eval_arc(0, requirepeak, checkaccel)
        clip.close()
        return code

    # ------------------------------------------------------------------
    #  step 5: YOLO body detection stays on the host (north_star)
    # ------------------------------------------------------------------

    def detect_juggler(self, display=False):
        """The reference runs a YOLOv2-tiny person detector through cv2.dnn
        here (reference tracker.py:1441-1779); that model is outside this
        build.  A host-side detector can be plugged in as ``body_detector``
        (a callable taking the scanner and returning {frame: (x, y, w, h,
        True)} in source-video pixels).  Without one, notes['body'] is left
        empty and step 6 estimates every body box from the tags -- the
        reference's own fallback for frames its detector missed."""
        notes = self.notes
        if self.body_detector is not None:
            self._say(1, 'Juggler detection starting...')
            notes['body'] = dict(self.body_detector(self))
            notes['body_source'] = 'detector'
            self._say(1, 'Juggler detection done: {} frames'.format(
                len(notes['body'])))
        else:
            # not the reference's result: say so in the notes and, once per
            # process, to the caller whatever the verbosity (worker.py runs
            # at verbosity 0)
            notes['body'] = dict()
            notes['body_source'] = 'estimated'
            import warnings
            warnings.warn(
                'hawkeye_amd: step 5 ran without a body detector; step 6 '
                'estimates every body box from the tags, so hand assignment '
                'and throw/catch results can differ from the reference\'s '
                'YOLO-backed notes (notes[\'body_source\'] = \'estimated\')',
                RuntimeWarning, stacklevel=2)
            self._say(1, 'Juggler detection skipped (no body detector): body '
                         'positions will be estimated from the detected '
                         'objects\n')

    # ------------------------------------------------------------------
    #  step 6 (reference tracker.py:1785-2504): host post-processing
    # ------------------------------------------------------------------

    def analyze_juggling(self):
        from . import juggling
        if 'body' not in self.notes:
            self.notes['body'] = dict()
        juggling.analyze_juggling(self.notes, say=self._say,
                                  progress=(self._progress
                                            if getattr(self, '_callback', None)
                                            is not None else None))

    # ------------------------------------------------------------------
    #  statics (reference tracker.py:2510-2607)
    # ------------------------------------------------------------------

    def default_scanner_params():
        """Scanner constants, same keys and values as the reference's
        default_scanner_params() (they are stored in notes and read by the
        GUI)."""
        return dict(
            body_averaging_time_window_secs=0.2,
            min_blob_area_high_res=7.0, min_blob_area_low_res=1.0,
            max_blob_area_high_res=1000.0, max_blob_area_low_res=150.0,
            max_frame_height_cm=1000, default_frame_height_cm=300,
            max_fps=60, default_fps=30,
            sigmasq=15.0,
            max_frame_gap_in_arc=2,
            max_distance_pixels_480=5,
            g_window=0.25,
            radius_window_high_res=0.65, radius_window_low_res=0.75,
            min_tags_per_arc_high_fps=10, min_tags_per_arc_low_fps=5,
            min_tags_to_curve_fit=4,
        )

    def read_notes(filename):
        with open(filename, 'rb') as handle:
            return pickle.load(handle)

    def write_notes(notes, filename):
        path = os.path.abspath(filename)
        folder = os.path.dirname(path)
        if not os.path.exists(folder):
            os.makedirs(folder)
        if os.path.exists(path):
            os.remove(path)
        with open(path, 'wb') as handle:
            pickle.dump(notes, handle, protocol=pickle.HIGHEST_PROTOCOL)


def play_video(filename, notes=None, outfilename=None, startframe=0,
               keywait=False):
    """Debug player of the reference (reference tracker.py:2619-2756): an
    OpenCV highgui tool, outside the accelerated path."""
    raise ScannerException('play_video is a host-side debug UI and is not '
                           'part of the accelerated path')
