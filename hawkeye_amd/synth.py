"""
Deterministic synthetic juggling scenes.

The reference ships no test clips (its sample videos are external downloads,
reference README.md:25-27), so every test and benchmark input is generated
here from a seed.  A scene is a list of ideal parabolic ball flights in a
W x H frame; from it we derive either

* per-frame *disc lists* (ball and distractor centres/radii) that the device
  renderer in ``csrc/hk_render.hip`` turns into BGR frames (SURVEY.md §8d
  configurations B-E), or
* *tag sets* -- the step-2 output shape ``{frame: [(x, y, diameter), ...]}`` --
  with position noise, missed detections and uniform clutter, used to
  exercise steps 3-4 without running detection.

Everything is NumPy with the frozen ``RandomState`` bit stream, so the same
seed gives the same scene on every machine.
"""

import math

import numpy as np


G_CM_PER_S2 = 980.7          # same constant as reference tracker.py:552
FRAME_HEIGHT_CM = 300.0      # reference default_frame_height_cm


class Flight:
    """One ideal ball flight: throw at frame f0 from (x0, y_hand), catch at
    frame f1 at (x1, y_hand); y grows downwards."""

    __slots__ = ('f0', 'f1', 'x0', 'x1', 'y_hand', 'g')

    def __init__(self, f0, f1, x0, x1, y_hand, g):
        self.f0, self.f1, self.x0, self.x1 = f0, f1, x0, x1
        self.y_hand, self.g = y_hand, g

    @property
    def f_peak(self):
        return 0.5 * (self.f0 + self.f1)

    def position(self, frame):
        t = frame - self.f_peak
        half = 0.5 * (self.f1 - self.f0)
        x = self.x0 + (self.x1 - self.x0) * (frame - self.f0) / (self.f1 -
                                                                self.f0)
        y = self.y_hand - 0.5 * self.g * (half * half - t * t)
        return x, y


class JugglingScene:
    """A cascade-like pattern of ``balls`` objects filmed at ``fps``.

    Flight time is 0.16*balls seconds; throws alternate between two hands at
    a steady rhythm chosen so that on average ``balls - 0.7`` objects are in
    the air.  ``tilt_deg`` rotates the whole pattern about the frame centre
    (positive = counter-clockwise juggler, the reference's sign convention,
    reference tracker.py:1047-1052).
    """

    def __init__(self, width, height, fps, balls, frames, seed,
                 tilt_deg=0.0):
        self.width, self.height, self.fps = width, height, float(fps)
        self.balls, self.frames, self.seed = balls, frames, seed
        self.tilt = math.radians(tilt_deg)
        rng = np.random.RandomState(seed)

        cm_per_px = FRAME_HEIGHT_CM / height
        self.g = G_CM_PER_S2 / (cm_per_px * self.fps**2)    # px / frame^2
        air = 0.16 * balls * self.fps                       # frames
        beat = air / (balls - 0.7)
        y_hand = 0.90 * height
        # keep the apex inside the picture
        apex = y_hand - self.g * air * air / 8.0
        if apex < 0.04 * height:
            air = math.sqrt(8.0 * (y_hand - 0.04 * height) / self.g)
            beat = air / (balls - 0.7)
        xc = 0.5 * width
        spread = 0.11 * height

        self.flights = []
        f0 = -air * rng.uniform(0.2, 1.0)
        k = 0
        while f0 < frames:
            side = 1.0 if (k % 2 == 0) else -1.0
            x0 = xc + side * (spread * 0.45 + rng.normal(0, 0.02 * height))
            x1 = xc - side * (spread * 1.35 + rng.normal(0, 0.03 * height))
            dur = air * (1.0 + rng.normal(0, 0.03))
            yh = y_hand + rng.normal(0, 0.01 * height)
            self.flights.append(Flight(f0, f0 + dur, x0, x1, yh, self.g))
            f0 += beat * (1.0 + rng.normal(0, 0.02))
            k += 1

    def _to_screen(self, x, y):
        if self.tilt == 0.0:
            return x, y
        cx, cy = 0.5 * self.width, 0.5 * self.height
        s, c = math.sin(self.tilt), math.cos(self.tilt)
        dx, dy = x - cx, y - cy
        return cx + dx * c + dy * s, cy + dy * c - dx * s

    def ball_positions(self, frame):
        """Screen positions of every ball in the air at ``frame``."""
        out = []
        for fl in self.flights:
            if fl.f0 <= frame <= fl.f1:
                x, y = fl.position(frame)
                out.append(self._to_screen(x, y))
        return out


def make_tags(scene, seed, clutter_per_frame=6.0, pos_sigma=0.6,
              detect_prob=0.92, diameter=None, diameter_sigma=0.4):
    """Tag set in the shape step 2 produces: ``{frame: [(x, y, d), ...]}``.

    ``d`` is a blob *diameter* (that is what the reference stores in
    Balltag.radius, reference tracker.py:414-417).  Clutter tags are uniform
    over the frame with diameters in [0.3, 1.7] x the ball diameter.
    """
    rng = np.random.RandomState(seed)
    if diameter is None:
        diameter = scene.height / 45.0
    meas = {}
    for f in range(scene.frames):
        row = []
        for (x, y) in scene.ball_positions(f):
            if rng.uniform() > detect_prob:
                continue
            if not (0 <= x < scene.width and 0 <= y < scene.height):
                continue
            row.append((float(x + rng.normal(0, pos_sigma)),
                        float(y + rng.normal(0, pos_sigma)),
                        float(diameter + rng.normal(0, diameter_sigma))))
        for _ in range(rng.poisson(clutter_per_frame)):
            row.append((float(rng.uniform(0, scene.width)),
                        float(rng.uniform(0, scene.height)),
                        float(diameter * rng.uniform(0.3, 1.7))))
        # detection order inside a frame is arbitrary; shuffle so nothing
        # downstream can lean on "balls first"
        rng.shuffle(row)
        meas[f] = row
    return meas


def tags_to_soa(meas, frame_count=None):
    """Flatten ``{frame: [(x, y, d), ...]}`` into the SoA arrays the native
    library takes: int32 frame, float64 x, y, radius, total_weight, plus the
    CSR ``frame_start`` index (length frame_count + 1)."""
    if frame_count is None:
        frame_count = (max(meas) + 1) if meas else 0
    counts = [len(meas.get(f, ())) for f in range(frame_count)]
    n = int(sum(counts))
    frame = np.empty(n, np.int32)
    x = np.empty(n, np.float64)
    y = np.empty(n, np.float64)
    r = np.empty(n, np.float64)
    start = np.zeros(frame_count + 1, np.int32)
    k = 0
    for f in range(frame_count):
        start[f] = k
        for (tx, ty, td) in meas.get(f, ()):
            frame[k], x[k], y[k], r[k] = f, tx, ty, td
            k += 1
    start[frame_count] = k
    return frame, x, y, r, np.ones(n, np.float64), start


# ---------------------------------------------------------------------------
# rendered clips (SURVEY.md §8d configurations)
# ---------------------------------------------------------------------------

CLIPS = {
    # name: width, height, fps, balls, frames, ball radius, noise sigma,
    #       drift amplitude / Hz, distractors, seed
    'B_720p30_5ball': dict(width=1280, height=720, fps=30.0, balls=5,
                           frames=900, ball_r=8.0, sigma=2.0, drift=0.0,
                           drift_hz=0.0, distractors=0, seed=1001),
    'C_1080p60_9ball_noisy': dict(width=1920, height=1080, fps=60.0, balls=9,
                                  frames=1800, ball_r=10.0, sigma=4.0,
                                  drift=8.0, drift_hz=0.2, distractors=20,
                                  seed=1002),
    'D_4k240_9ball_long': dict(width=3840, height=2160, fps=240.0, balls=9,
                               frames=14400, ball_r=16.0, sigma=3.0,
                               drift=0.0, drift_hz=0.0, distractors=0,
                               seed=1003),
}


def clip_E(i):
    """Clip i of configuration E: 64 clips 1920x1080@60, 600 frames, 3..9
    balls cycling, seeds 2000+i."""
    return dict(width=1920, height=1080, fps=60.0, balls=3 + i % 7,
                frames=600, ball_r=10.0, sigma=3.0, drift=0.0, drift_hz=0.0,
                distractors=0, seed=2000 + i)


MAX_DISCS = 40


def disc_lists(spec, scene=None):
    """float32 [frames][MAX_DISCS][4] = (x, y, radius, kind) per frame:
    balls of the juggling scene (kind 0) plus random-walk distractor blobs
    with radii in [2, 14] (kinds 1-3)."""
    if scene is None:
        scene = JugglingScene(spec['width'], spec['height'], spec['fps'],
                              spec['balls'], spec['frames'], spec['seed'])
    F = spec['frames']
    out = np.zeros((F, MAX_DISCS, 4), np.float32)
    rng = np.random.RandomState(spec['seed'] + 7)
    nd = spec['distractors']
    if nd:
        pos = np.stack([rng.uniform(0, spec['width'], nd),
                        rng.uniform(0, spec['height'], nd)], 1)
        rad = rng.uniform(2.0, 14.0, nd)
        kind = 1 + rng.randint(0, 3, nd)
        vel = rng.normal(0, 1.0, (nd, 2))
    for f in range(F):
        k = 0
        for (x, y) in scene.ball_positions(f):
            if k < MAX_DISCS - nd:
                out[f, k] = (x, y, spec['ball_r'], 0)
                k += 1
        if nd:
            vel = 0.95 * vel + rng.normal(0, 0.35, (nd, 2))
            pos = pos + vel
            pos[:, 0] %= spec['width']
            pos[:, 1] %= spec['height']
            out[f, MAX_DISCS - nd:, 0:2] = pos
            out[f, MAX_DISCS - nd:, 2] = rad
            out[f, MAX_DISCS - nd:, 3] = kind
    return out


class ClipRenderer:
    """Renders frames of one synthetic clip on the device (uint8 BGR)."""

    def __init__(self, spec, device=0, juggler=True):
        import torch
        from . import _native as nat
        self._nat, self._torch = nat, torch
        self.spec = dict(spec)
        self.device = torch.device('cuda', device)
        self.scene = JugglingScene(spec['width'], spec['height'], spec['fps'],
                                   spec['balls'], spec['frames'], spec['seed'])
        self.discs = torch.from_numpy(disc_lists(spec, self.scene)).to(
            self.device)
        self.juggler = 1 if juggler else 0

    def render(self, first, count, out=None):
        """Frames [first, first+count) as a uint8 tensor [count][H][W][3]."""
        torch, nat, s = self._torch, self._nat, self.spec
        if first < 0 or first + count > s['frames']:
            raise ValueError('frame range outside the clip')
        if out is None:
            out = torch.empty((count, s['height'], s['width'], 3),
                              dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        done = 0
        while done < count:         # grid.z limit
            n = min(count - done, 4096)
            nat.check(nat.lib.hk_render_frames(
                s['width'], s['height'], s['fps'], s['seed'], s['sigma'],
                s['drift'], s['drift_hz'], self.juggler,
                self.discs.data_ptr(), MAX_DISCS, first + done, n,
                out[done].data_ptr(), out.stride(0), out.stride(1), stream))
            done += n
        return out
