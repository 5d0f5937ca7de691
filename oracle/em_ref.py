"""
ORACLE -- test infrastructure only.  Never imported by the product
(hawkeye_amd/); only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use it.

CPU restatement of the reference's steps 3-4 (preliminary arcs + EM) on flat
arrays.  Tags are integer indices into SoA arrays, in (frame, position in the
frame's list) order; arcs are small records holding a *sorted list* of tag
indices.  Where the reference iterates a Python ``set`` of objects (order =
memory addresses, i.e. not reproducible) this
restatement iterates in ascending tag index and breaks exact ties towards the
lower index, so it is deterministic; it therefore agrees with the reference to
rounding (<= 1e-9 relative on the golden runs), not to the bit.

Pinned by tests/golden/*.json, which were produced by running the reference
itself (tests/golden/make_golden.py).

Reference call sites restated here (all in /root/reference/hawkeye/):
  Arc.position / distance / frame_range   types.py:59-102
  fit (M-step)                             tracker.py:698-798
  estimate_tilt                            tracker.py:1041-1124
  eval_arc / accel_ok                      tracker.py:800-878
  neighbor lists / construct_arcs          tracker.py:487-696
  find_global_params / build_initial_arcs  tracker.py:450-485, 880-906
  e_step (calculate_weights)               tracker.py:992-1039
  merge_pass / prune_pass                  tracker.py:1126-1329
  clean / em_optimize                      tracker.py:912-990, 1331-1435
"""

import math
from statistics import median

import numpy as np

# The reference writes t**3, t**4 and T2**3 (tracker.py:729-730, 757), i.e.
# pow(); the native code multiplies.  The two differ by at most one ulp, far
# inside the reference's own run-to-run variation (it sums in set order), but
# while a trial arc's peak lies hundreds of frames from its tags the fit is
# ill-conditioned enough to turn that ulp into a different arc set.  The
# randomised comparison (tests/fuzz_arcs.py) sets this to take that noise out.
POWERS_AS_PRODUCTS = False

G_CM = 980.7


def default_params():
    """Scanner constants (reference tracker.py:2510-2570); only the keys the
    hot path reads."""
    return {
        'default_frame_height_cm': 300, 'max_frame_height_cm': 1000,
        'default_fps': 30, 'max_fps': 60, 'sigmasq': 15.0,
        'max_frame_gap_in_arc': 2, 'max_distance_pixels_480': 5,
        'g_window': 0.25, 'radius_window_high_res': 0.65,
        'radius_window_low_res': 0.75, 'min_tags_per_arc_high_fps': 10,
        'min_tags_per_arc_low_fps': 5, 'min_tags_to_curve_fit': 4,
    }


class Arc:
    __slots__ = ('a', 'b', 'c', 'e', 'f_peak', 'tags', 'id', 'done')

    def __init__(self):
        self.a = self.b = self.c = self.e = self.f_peak = 0.0
        self.tags = []          # sorted tag indices
        self.id = None
        self.done = False

    def clone(self):
        o = Arc()
        o.a, o.b, o.c, o.e, o.f_peak = self.a, self.b, self.c, self.e, \
            self.f_peak
        o.tags, o.id, o.done = self.tags, self.id, self.done
        return o

    def params(self):
        return (self.a, self.b, self.c, self.e, self.f_peak)


class Scene:
    """All state of one clip between step 2 and the end of step 4."""

    def __init__(self, meas, width, height, fps, frame_count=None,
                 params=None):
        """``meas`` = {frame: [(x, y, diameter), ...]} (step-2 output)."""
        if frame_count is None:
            frame_count = (max(meas) + 1) if meas else 0
        self.W, self.H, self.fps, self.F = width, height, fps, frame_count
        p = dict(default_params() if params is None else params)
        # derived keys, reference tracker.py:344-355 (scan video == source)
        p['min_tags_per_arc'] = (p['min_tags_per_arc_high_fps'] if fps >= 29
                                 else p['min_tags_per_arc_low_fps'])
        p['max_distance_pixels'] = p['max_distance_pixels_480'] * height / 480
        p['radius_window'] = (p['radius_window_high_res'] if height >= 480
                              else p['radius_window_low_res'])
        self.p = p
        self.maxd = p['max_distance_pixels']

        fr, xs, ys, rs, key = [], [], [], [], []
        self.start = [0] * (frame_count + 1)
        for f in range(frame_count):
            self.start[f] = len(fr)
            for k, (x, y, d) in enumerate(meas.get(f, ())):
                fr.append(f); xs.append(x); ys.append(y); rs.append(d)
                key.append((f, k))
        self.start[frame_count] = len(fr)
        self.n = len(fr)
        self.frame, self.x, self.y, self.r = fr, xs, ys, rs
        self.key = key                      # (frame, index in input frame)
        self.tw = [1.0] * self.n            # total_weight
        self.alive = [True] * self.n        # cleared by clean()
        self.tilt = 0.0
        self.g = None                       # g_px_per_frame_sq
        self.cm_per_pixel = None
        self.arcs = []
        # E-step output: per tag an insertion-ordered {arc: weight}
        self.w = [None] * self.n
        self.final_arc = [None] * self.n

    # -- arc geometry --------------------------------------------------------

    def position(self, arc, frame):
        t = frame - arc.f_peak
        xj = arc.a + arc.b * t
        yj = arc.c + arc.e * t * t
        s, c = math.sin(self.tilt), math.cos(self.tilt)
        return xj * c + yj * s, yj * c - xj * s

    def dist(self, arc, i):
        px, py = self.position(arc, self.frame[i])
        return math.sqrt((px - self.x[i])**2 + (py - self.y[i])**2)

    def frame_range(self, arc):
        if arc.b == 0:
            lo_x, hi_x = 0, self.F
        else:
            lo_x = arc.f_peak - arc.a / arc.b
            hi_x = arc.f_peak + (self.W - arc.a) / arc.b
        if lo_x > hi_x:
            lo_x, hi_x = hi_x, lo_x
        lo_x, hi_x = int(lo_x + 1), int(hi_x + 1)
        q = (self.H - arc.c) / arc.e
        lo_y = 0 if q < 0 else arc.f_peak - math.sqrt(q)
        hi_y = self.F if arc.e < 0 else 2 * arc.f_peak - lo_y
        if lo_y > hi_y:
            lo_y, hi_y = hi_y, lo_y
        lo_y, hi_y = int(lo_y + 1), int(hi_y + 1)
        return max(0, lo_x, lo_y), min(self.F - 1, hi_x, hi_y)

    def median_radius(self, arc):
        return median(self.r[i] for i in arc.tags)

    # -- least squares -------------------------------------------------------

    def _moments(self, arc, weight_of):
        T0 = T1 = T2 = T3 = T4 = 0.0
        X1 = TX = T2X = Y1 = TY = T2Y = 0.0
        for i in arc.tags:
            t = self.frame[i] - arc.f_peak
            w = weight_of(i)
            x, y = self.x[i], self.y[i]
            T0 += w; T1 += w * t; T2 += w * t**2
            if POWERS_AS_PRODUCTS:
                t2 = t * t
                T3 += w * (t2 * t); T4 += w * (t2 * t2)
            else:
                T3 += w * t**3; T4 += w * t**4
            X1 += w * x; TX += w * t * x; T2X += w * t**2 * x
            Y1 += w * y; TY += w * t * y; T2Y += w * t**2 * y
        return T0, T1, T2, T3, T4, X1, TX, T2X, Y1, TY, T2Y

    def fit(self, arc, weight_of):
        """Weighted parabola fit of one arc; leaves the arc untouched when the
        normal equations are (near-)singular."""
        if len(arc.tags) < 3:
            return
        T0, T1, T2, T3, T4, X1, TX, T2X, Y1, TY, T2Y = \
            self._moments(arc, weight_of)
        s, c = math.sin(self.tilt), math.cos(self.tilt)
        det2 = T0 * T2 - T1**2
        det3 = 2*T1*T2*T3 + T0*T2*T4 - T0*T3**2 - T1**2*T4 - \
            (T2 * T2 * T2 if POWERS_AS_PRODUCTS else T2**3)
        if abs(det2) < 1e-3 or abs(det3) < 1e-3:
            return
        bx1 = c * TX - s * TY
        bx2 = c * X1 - s * Y1
        b_new = (T0 * bx1 - T1 * bx2) / det2
        a_new = (-T1 * bx1 + T2 * bx2) / det2
        m11 = T0 * T2 - T1**2
        m12 = T1 * T2 - T0 * T3
        m13 = T1 * T3 - T2**2
        m22 = T0 * T4 - T2**2
        m23 = T2 * T3 - T1 * T4
        m33 = T2 * T4 - T3**2
        by1 = c * T2Y + s * T2X
        by2 = c * TY + s * TX
        by3 = c * Y1 + s * X1
        e_new = (m11 * by1 + m12 * by2 + m13 * by3) / det3
        d_new = (m12 * by1 + m22 * by2 + m23 * by3) / det3
        c_new = (m13 * by1 + m23 * by2 + m33 * by3) / det3
        if any(math.isnan(v) for v in (a_new, b_new, c_new, d_new, e_new)) \
                or e_new == 0:
            return
        shift = -d_new / (2.0 * e_new)
        arc.f_peak += shift
        arc.a = a_new + b_new * shift
        arc.c = c_new + (d_new * shift + e_new * shift**2)
        arc.b = b_new
        arc.e = e_new

    def _w_em(self, arc):
        """Weight lookup used by fits over the arc list: E-step weight if the
        tag carries weights, else 1.0 (reference tracker.py:722-725)."""
        def look(i):
            wi = self.w[i]
            return 1.0 if wi is None else wi[arc]
        return look

    def fit_all(self):
        for arc in self.arcs:
            self.fit(arc, self._w_em(arc))

    def estimate_tilt(self):
        total, count = 0.0, 0
        for arc in self.arcs:
            if len(arc.tags) < 3:
                continue
            T0, T1, T2, T3, T4, X1, TX, T2X, Y1, TY, T2Y = \
                self._moments(arc, self._w_em(arc))
            det3 = 2*T1*T2*T3 + T0*T2*T4 - T0*T3**2 - T1**2*T4 - T2**3
            if abs(det3) < 1e-3:
                continue
            m11 = T0 * T2 - T1**2
            m12 = T1 * T2 - T0 * T3
            m13 = T1 * T3 - T2**2
            e_y = (m11 * T2Y + m12 * TY + m13 * Y1) / det3
            e_x = (m11 * T2X + m12 * TX + m13 * X1) / det3
            if self.accel_ok(e_y):
                total += math.atan(e_x / e_y)
                count += 1
        self.tilt = total / count if count else 0.0

    # -- arc quality ---------------------------------------------------------

    def accel_ok(self, e):
        if e < 0:
            return False
        if self.g is not None:
            gw = self.p['g_window']
            return not (2 * e < (1 - gw) * self.g or
                        2 * e > (1 + gw) * self.g)
        max_cm_per_px = self.p['max_frame_height_cm'] / self.H
        return not (2 * e < G_CM / (max_cm_per_px * self.fps**2))

    def eval_arc(self, arc, requirepeak=False, checkaccel=True):
        if any(math.isnan(v) for v in arc.params()):
            return 1
        if not arc.tags:
            return 5
        if requirepeak:
            lo = min(self.frame[i] for i in arc.tags)
            hi = max(self.frame[i] for i in arc.tags)
            if not (lo <= arc.f_peak <= hi):
                return 3
        if checkaccel and not self.accel_ok(arc.e):
            return 4
        close = sum(1 for i in arc.tags if self.dist(arc, i) < self.maxd)
        return 2 if close < self.p['min_tags_per_arc'] else 0

    # -- step 3 --------------------------------------------------------------

    def neighbor_lists(self):
        gap = self.p['max_frame_gap_in_arc']
        v_max = (math.sqrt(2 * self.g * self.H) if self.g is not None
                 else None)
        nb = []
        for i in range(self.n):
            f = self.frame[i]
            lo, hi = max(0, f - gap - 1), min(self.F, f + gap + 2)
            cand = [j for j in range(self.start[lo], self.start[hi])
                    if self.frame[j] != f]
            vel = [math.sqrt((self.x[j] - self.x[i])**2 +
                             (self.y[j] - self.y[i])**2) /
                   abs(self.frame[j] - f) for j in cand]
            order = sorted(range(len(cand)), key=vel.__getitem__)   # stable
            if v_max is not None:
                order = [k for k in order if vel[k] <= v_max]
            nb.append([cand[k] for k in order])
        return nb

    def construct_arcs(self, maxcount=None):
        nb = self.neighbor_lists()
        maxd = self.maxd
        queue = sorted(range(self.n), key=self.y.__getitem__)       # stable
        done = [False] * self.n
        unit = lambda i: 1.0
        if self.g is not None:
            e0 = 0.5 * self.g
        else:
            e0 = 0.5 * G_CM / ((self.p['default_frame_height_cm'] / self.H) *
                               self.fps**2)
        out = []
        for t1 in queue:
            if done[t1]:
                continue
            made = False
            for t2 in nb[t1]:
                if done[t2]:
                    continue
                arc = Arc()
                members = {t1, t2}
                taken = {self.frame[t1], self.frame[t2]}
                frontier = {j for j in set(nb[t1]) | set(nb[t2])
                            if self.frame[j] not in taken and not done[j]}
                arc.e = e0
                f1, f2 = self.frame[t1], self.frame[t2]
                arc.f_peak = ((self.y[t1] - self.y[t2]) /
                              (2.0 * arc.e * (f2 - f1)) + 0.5 * (f1 + f2))
                arc.c = self.y[t1] - arc.e * (f1 - arc.f_peak)**2
                arc.b = (self.x[t1] - self.x[t2]) / (f1 - f2)
                arc.a = self.x[t1] - arc.b * (f1 - arc.f_peak)
                arc.tags = sorted(members)
                bad = False
                while frontier:
                    best, best_d = None, None
                    for j in sorted(frontier):
                        d = self.dist(arc, j)
                        if best is None or d < best_d:
                            best, best_d = j, d
                    if best_d > maxd:
                        break
                    members.add(best)
                    arc.tags = sorted(members)
                    taken.add(self.frame[best])
                    frontier |= set(nb[best])
                    frontier = {j for j in frontier
                                if self.frame[j] not in taken and not done[j]}
                    if len(members) > self.p['min_tags_to_curve_fit']:
                        self.fit(arc, unit)
                    if (math.isnan(arc.e) or arc.e <= 0 or
                            any(self.dist(arc, j) > maxd for j in arc.tags)):
                        bad = True
                        break
                if not bad and self.eval_arc(arc, requirepeak=False) == 0:
                    for j in arc.tags:
                        done[j] = True
                    out.append(arc)
                    made = True
                if made:
                    break
            if maxcount is not None and len(out) >= maxcount:
                break
            done[t1] = True
        return out

    def find_global_params(self, arcs):
        if not arcs:
            return
        top = sorted(arcs, key=lambda a: len(a.tags), reverse=True)[:10]
        self.g = 2 * median(a.e for a in top)
        self.cm_per_pixel = G_CM / (self.g * self.fps**2)

    def build_initial_arcs(self):
        self.find_global_params(self.construct_arcs(maxcount=5))
        arcs = self.construct_arcs()
        self.find_global_params(arcs)
        arcs.sort(key=lambda a: a.f_peak)
        for k, arc in enumerate(arcs, start=1):
            arc.id = k
        self.arcs = arcs
        # tags of accepted arcs carry weight 1.0 into step 4; modelled by
        # w[i] is None  ->  1.0 in _w_em()
        self.w = [None] * self.n

    # -- step 4 --------------------------------------------------------------

    def e_step(self):
        rw = self.p['radius_window']
        sig = self.p['sigmasq']
        for i in range(self.n):
            if self.alive[i]:
                self.w[i] = {}
        for arc in self.arcs:
            m = self.median_radius(arc)
            r_lo, r_hi = m * (1 - rw), m * (1 + rw)
            arc.tags = []
            lo, hi = self.frame_range(arc)
            for f in range(lo, hi):
                px, py = self.position(arc, f)
                for i in range(self.start[f], self.start[f + 1]):
                    if not self.alive[i] or not (r_lo <= self.r[i] <= r_hi):
                        continue
                    q = ((px - self.x[i])**2 + (py - self.y[i])**2) / sig
                    if q < 5.0:
                        self.w[i][arc] = math.exp(-q)
        for i in range(self.n):
            if not self.alive[i]:
                continue
            wi = self.w[i]
            total = sum(wi.values())
            if total > 1e-5:
                for arc in wi:
                    wi[arc] = wi[arc] * self.tw[i] / total
                    arc.tags.append(i)
        # arcs were visited in list order per tag, tags in index order, so
        # every arc.tags is already sorted

    def _good_tags(self, arc):
        rw = self.p['radius_window']
        m = self.median_radius(arc)
        r_lo, r_hi = m * (1 - rw), m * (1 + rw)
        return [i for i in arc.tags
                if self.dist(arc, i) < self.maxd and self.tw[i] > 0.95
                and r_lo <= self.r[i] <= r_hi]

    def merge_pass(self):
        """One reference merge_arcs() call: True after the first merger."""
        arcs = self.arcs
        for a1 in arcs:
            if a1.done:
                continue
            if not a1.tags:
                a1.done = True
                continue
            lo1, hi1 = self.frame_range(a1)
            good1 = self._good_tags(a1)
            if len(good1) < 3:
                a1.done = True
                continue
            for a2 in arcs:
                if a2 is a1:
                    continue
                if not a2.tags:
                    a2.done = True
                    continue
                lo2, hi2 = self.frame_range(a2)
                if hi2 < lo1 or lo2 > hi1:
                    continue
                good2 = self._good_tags(a2)
                if len(good2) < 3:
                    continue
                trial = a1.clone()
                trial.tags = sorted(set(good1) | set(good2))
                if len(trial.tags) < self.p['min_tags_per_arc']:
                    continue
                self.fit(trial, self.tw.__getitem__)
                poor = sum(1 for i in trial.tags
                           if self.dist(trial, i) > self.maxd)
                if poor > 2:
                    continue
                arcs.remove(a2)
                a1.a, a1.b, a1.c, a1.e, a1.f_peak = trial.params()
                a1.tags = trial.tags
                return True
            a1.done = True
        return False

    def prune_pass(self):
        for arc in self.arcs:
            if self.eval_arc(arc, requirepeak=False) > 0:
                self.arcs.remove(arc)
                return True
        return False

    def clean(self):
        tag_done = [False] * self.n
        for arc in self.arcs:
            arc.done = False
        keep = set(self.arcs)
        again = True
        while again:
            for i in range(self.n):
                if not self.alive[i] or tag_done[i]:
                    continue
                live = [a for a in self.w[i] if a in keep]
                if any(self.dist(a, i) < self.maxd for a in live):
                    tag_done[i] = True
                    continue
                self.alive[i] = False
                for a in live:
                    a.tags.remove(i)
                    a.done = False
            again = False
            doomed = []
            for arc in self.arcs:
                if arc.done:
                    continue
                if self.eval_arc(arc, requirepeak=True) > 0:
                    doomed.append(arc)
                else:
                    arc.done = True
            for arc in doomed:
                self.arcs.remove(arc)
                keep.discard(arc)
                for i in arc.tags:
                    tag_done[i] = False
                again = True
        for arc in self.arcs:
            arc.tags = []
        for i in range(self.n):
            if not self.alive[i]:
                continue
            best, best_d = None, None
            for a in self.w[i]:
                if a not in keep:
                    continue
                d = self.dist(a, i)
                if best is None or d < best_d:
                    best, best_d = a, d
            self.final_arc[i] = best
            best.tags.append(i)
            self.w[i] = None        # -> unit weights in the closing fit

    def em_optimize(self):
        again = True
        while again:
            self.estimate_tilt()
            self.fit_all()
            again = False
            for arc in self.arcs:
                arc.done = False
            while self.merge_pass():
                again = True
            while self.prune_pass():
                again = True
            self.e_step()
            self.fit_all()
        self.clean()
        self.estimate_tilt()
        self.fit_all()
        self.arcs.sort(key=lambda a: a.f_peak)
