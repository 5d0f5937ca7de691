"""
Event and arc record types of the scanner hot path.

Drop-in for the reference's ``hawkeye/types.py`` (``Balltag`` reference
types.py:11-24, ``Ballarc`` reference types.py:27-234): same class names, same
constructor signatures, same attribute sets and same method names, so that
``notes`` dictionaries produced here are consumed unchanged by the reference's
GUI/worker code.  Both classes advertise ``__module__ = 'hawkeye.types'`` so a
pickled ``notes`` file written by either implementation loads in the other
(reference tracker.py:2572-2607); the repo-root ``hawkeye/`` package re-exports
them under that name.

The arithmetic here is the *host-side mirror* used when Python code touches an
individual arc (GUI overlays, tests).  The hot loops never call these methods:
steps 2-4 run in the native library on SoA copies of the same fields.
"""

import math
import statistics


class Balltag:
    """One detection event: a blob seen at (x, y) in frame ``frame``.

    ``radius`` holds whatever step 2 stored there, which is the blob
    *diameter* scaled to source-video pixels (reference tracker.py:414-417).
    """

    def __init__(self, frame, x, y, radius, total_weight=1.0):
        self.frame = frame
        self.x = x
        self.y = y
        self.radius = radius
        self.total_weight = total_weight
        self.arc = None

    def __str__(self):
        return '{}: {}'.format(self.__class__, self.__dict__)


class Ballarc:
    """A parabola x = a + b t, y = c + e t^2 with t = frame - f_peak, in
    juggler coordinates (screen coordinates rotated by notes['camera_tilt']).
    """

    # attributes filled in by the (out-of-scope) step 6, kept so that the
    # attribute set matches reference types.py:43-57
    _STEP6_FIELDS = (
        'f_throw', 'x_throw', 'f_catch', 'x_catch', 'height', 'run_id',
        'throw_id', 'hand_throw', 'hand_catch', 'next', 'prev',
        'throw_error_s', 'throw_error_cm', 'catch_error_s', 'catch_error_cm')

    def __init__(self):
        self.f_peak = 0.0
        self.a = 0.0
        self.b = 0.0
        self.c = 0.0
        self.e = 0.0
        self.tags = set()
        self.id_ = None
        for name in Ballarc._STEP6_FIELDS:
            setattr(self, name, None)

    # -- geometry ----------------------------------------------------------

    def get_position(self, frame, notes):
        """Screen position at ``frame`` (reference types.py:59-68)."""
        t = frame - self.f_peak
        xj = self.a + self.b * t
        yj = self.c + self.e * t * t
        tilt = notes['camera_tilt']
        sn, cs = math.sin(tilt), math.cos(tilt)
        return (xj * cs + yj * sn, yj * cs - xj * sn)

    def get_distance_from_tag(self, tag, notes):
        px, py = self.get_position(tag.frame, notes)
        return math.sqrt((px - tag.x)**2 + (py - tag.y)**2)

    def get_distance_from_arc(self, arc2, frame, notes):
        px, py = self.get_position(frame, notes)
        qx, qy = arc2.get_position(frame, notes)
        return math.sqrt((px - qx)**2 + (py - qy)**2)

    def get_frame_range(self, notes):
        """Frames during which the arc is inside the picture (reference
        types.py:79-102).  Quirks kept on purpose: juggler-coordinate a, b, c
        are compared against screen width/height, both ends go through
        ``int(v + 1)``, and e == 0 raises ZeroDivisionError.
        """
        w = notes['frame_width']
        h = notes['frame_height']
        n = notes['frame_count']

        if self.b == 0:
            x_lo, x_hi = 0, n
        else:
            x_lo = self.f_peak - self.a / self.b
            x_hi = self.f_peak + (w - self.a) / self.b
        x_lo, x_hi = min(x_lo, x_hi), max(x_lo, x_hi)
        x_lo, x_hi = int(x_lo + 1), int(x_hi + 1)

        drop = (h - self.c) / self.e
        y_lo = 0 if drop < 0 else self.f_peak - math.sqrt(drop)
        y_hi = n if self.e < 0 else 2 * self.f_peak - y_lo
        y_lo, y_hi = min(y_lo, y_hi), max(y_lo, y_hi)
        y_lo, y_hi = int(y_lo + 1), int(y_hi + 1)

        return max(0, x_lo, y_lo), min(n - 1, x_hi, y_hi)

    def get_tag_range(self):
        frames = [t.frame for t in self.tags]
        return min(frames), max(frames)

    def get_median_tag_radius(self):
        return statistics.median(t.radius for t in self.tags)

    # -- arc/arc closest approach (GUI-era helper, no hot-path caller) -------

    def closest_approach_frame(self, arc2, notes):
        """Frame at which this arc and ``arc2`` are nearest to each other
        within their common lifetime, or None if the lifetimes are disjoint
        (reference types.py:113-155).
        """
        def lifetime(arc):
            if arc.f_throw is not None and arc.f_catch is not None:
                return arc.f_throw, arc.f_catch
            return arc.get_frame_range(notes)

        s1, e1_ = lifetime(self)
        s2, e2_ = lifetime(arc2)
        if s1 > e2_ or s2 > e1_:
            return None
        lo, hi = max(s1, s2), min(e1_, e2_)

        # d/df |p1(f) - p2(f)|^2 = 0 written in u = f - f_peak(self)
        da, db = self.a - arc2.a, self.b - arc2.b
        dc, de = self.c - arc2.c, self.e - arc2.e
        dp = self.f_peak - arc2.f_peak
        e2 = arc2.e
        k3 = 2 * de**2
        k2 = -6 * e2 * de * dp
        k1 = db**2 + 2 * dc * de - 2 * e2 * (self.e - 3 * e2) * dp**2
        k0 = da * db + dp * (arc2.b * (-db) + 2 * e2 * (-dc + e2 * dp**2))

        cands = [u + self.f_peak for u in Ballarc.solve_cubic(k3, k2, k1, k0)]
        cands = [f for f in cands if lo < f < hi] + [lo, hi]
        return min(cands,
                   key=lambda f: self.get_distance_from_arc(arc2, f, notes))

    def solve_cubic(A, B, C, D):
        """Real roots of A x^3 + B x^2 + C x + D = 0 (degenerate orders
        included), as a list (reference types.py:157-234).
        """
        def cbrt(v):
            return math.copysign(abs(v) ** (1.0 / 3.0), v)

        if A == 0:
            if B == 0:
                return [-D / C] if C != 0 else []
            disc = C * C - 4 * B * D
            if disc < 0:
                return []
            r = math.sqrt(disc)
            return [(-C + r) / (2 * B), (-C - r) / (2 * B)]

        # depressed cubic y^3 + p y + q = 0 with x = y + shift
        shift = -B / (3.0 * A)
        p = (3 * C / A - (B / A)**2) / 3
        q = (2 * (B / A)**3 - 9 * B * C / A**2 + 27 * D / A) / 27
        disc = q * q / 4 + p**3 / 27

        if p == 0 and q == 0 and disc == 0:
            roots = [-cbrt(D / A)]
        elif disc <= 0:
            rho = math.sqrt(q * q / 4 - disc)
            phi = math.acos(-(q / (2 * rho)))
            m = rho ** (1.0 / 3.0)
            cs, sn = math.cos(phi / 3.0), math.sqrt(3) * math.sin(phi / 3.0)
            roots = [2 * m * cs + shift, -m * (cs + sn) + shift,
                     -m * (cs - sn) + shift]
        else:
            sq = math.sqrt(disc)
            roots = [cbrt(-q / 2 + sq) + cbrt(-q / 2 - sq) + shift]

        for x in roots:
            resid = ((A * x + B) * x + C) * x + D
            if abs(resid) > 1e-2:
                print(f'ERROR: f({A}, {B}, {C}, {D}; {x}) = {resid}')
        return roots


# Pickle identity shared with the reference implementation.
Balltag.__module__ = 'hawkeye.types'
Ballarc.__module__ = 'hawkeye.types'
