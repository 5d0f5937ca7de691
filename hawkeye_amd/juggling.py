"""
Step 6 of the scanner: from the fitted arcs to a description of the juggling
(reference tracker.py:1785-2504, ``VideoScanner.analyze_juggling`` and its
helpers).  SURVEY.md §8(f) row N3: cheap host-side post-processing of the
final arcs, needed so that ``process(steps=(..., 6))`` hands the caller the
same ``notes`` keys and ``Ballarc`` attributes the reference does
(``widgets.py`` / ``mainwindow.py`` read them).

Written from the behaviour of the reference, checked against golden vectors
produced by running the reference itself (tests/golden/make_golden.py,
``step6`` blocks).  Pure Python on ``notes`` objects, like the reference; no
GPU work: a clip has ~10^2 arcs.

What it adds
  notes['origin']   frame -> (x, y) of the hands' rest point on the body axis
  notes['body']     missing frames filled with an estimated (x, y, w, h, False)
  notes['runs'], notes['run']   number of runs and one dictionary per run
  per arc           f_throw, f_catch, x_throw, x_catch, height, run_id,
                    throw_id, hand_throw, hand_catch, prev, next and the
                    timing errors of analyze_run_form
and what it removes: tags below hand level, arcs that do not survive that,
arcs in runs of a single throw.

Step 5 (YOLO body detection) is outside this build; ``notes['body']`` may be
empty, in which case every body box is estimated from the tags, which is the
reference's own fallback for frames the detector missed.
"""
from math import ceil, cos, floor, isnan, sin, sqrt
from statistics import mean, median

G_CM_PER_S2 = 980.7

# ideal scoop as a fraction of the pattern width, by ball count (reference
# tracker.py:2425-2428: an analytical collision-probability model)
SCOOP_OVER_WIDTH = (0.0, 0.5, 0.5, 0.4, 0.4, 0.38, 0.45, 0.32, 0.45, 0.26,
                    0.45, 0.23, 0.45, 0.19, 0.45)


def acceleration_is_plausible(notes, e):
    """reference tracker.py:842-878"""
    if e < 0:
        return False
    sp = notes['scanner_params']
    if 'g_px_per_frame_sq' in notes:
        g = notes['g_px_per_frame_sq']
        return not (2 * e < (1 - sp['g_window']) * g or
                    2 * e > (1 + sp['g_window']) * g)
    cm_per_px_max = sp['max_frame_height_cm'] / notes['frame_height']
    fps = notes['fps'] if 'fps' in notes else sp['max_fps']
    return not (2 * e) < G_CM_PER_S2 / (cm_per_px_max * fps**2)


def arc_failure_code(notes, arc, requirepeak=False, checkaccel=True):
    """eval_arc (reference tracker.py:800-840): 0 = keep, else why not."""
    if any(isnan(v) for v in (arc.a, arc.b, arc.c, arc.e, arc.f_peak)):
        return 1
    if not arc.tags:
        return 5
    if requirepeak:
        first, last = arc.get_tag_range()
        if not first <= arc.f_peak <= last:
            return 3
    if checkaccel and not acceleration_is_plausible(notes, arc.e):
        return 4
    sp = notes['scanner_params']
    near = sum(1 for t in arc.tags
               if arc.get_distance_from_tag(t, notes) <
               sp['max_distance_pixels'])
    return 2 if near < sp['min_tags_per_arc'] else 0


# ---------------------------------------------------------------------------
#  body origin, tags below the hands
# ---------------------------------------------------------------------------

def set_body_origin(notes, say=lambda level, text: None):
    """notes['origin'][frame] for every frame some arc is in flight
    (reference tracker.py:1867-1946).  Frames without a body measurement get
    an estimated box from the outermost tags within +-120 frames."""
    origin = notes['origin'] = dict()
    say(2, 'setting hand levels...')
    nframes = notes['frame_count']
    in_flight = [0] * nframes
    for arc in notes['arcs']:
        first, last = arc.get_frame_range(notes)
        for f in range(first, last + 1):
            in_flight[f] += 1

    body = notes['body']
    measured = None             # last box that came from the detector
    filled = 0
    for f in range(nframes):
        if not in_flight[f]:
            continue
        if f in body:
            measured = body[f]
        else:
            tags = []
            for g in range(max(f - 120, 0), min(f + 120, nframes)):
                tags.extend(notes['meas'][g])
            if tags:
                by_x = sorted(tags, key=lambda t: t.x)
                left = median(t.x for t in by_x[:5])
                right = median(t.x for t in by_x[-5:])
                if measured is not None:
                    _, top, w, h, _ = measured
                    x = 0.5 * (left + right - w)
                else:
                    by_y = sorted(tags, key=lambda t: t.y)
                    bottom = median(t.y for t in by_y[-5:])
                    w = 0.7 * (right - left)
                    h = 0.8 * w
                    x, top = 0.5 * (left + right - w), bottom - h
                body[f] = (x, top, w, h, False)
                filled += 1
            elif measured is not None:
                body[f] = measured
                filled += 1
            else:
                say(2, '   problem adding body location to frame {}'.format(f))
        x, top, w, _, _ = body[f]
        # hands rest about 50 cm below the top of the head
        origin[f] = (x + 0.5 * w, top + 50.0 / notes['cm_per_pixel'])
    if filled:
        say(2, '  added missing body measurements to {} frames'.format(filled))


def remove_tags_below_hands(notes, say=lambda level, text: None):
    """reference tracker.py:1948-1996"""
    say(2, 'removing detections below hand level...')
    doomed = []
    tags_removed = 0
    for arc in notes['arcs']:
        hand_level = notes['origin'][round(arc.f_peak)][1]
        low = [t for t in arc.tags if t.y > hand_level]
        for t in low:
            arc.tags.remove(t)
            notes['meas'][t.frame].remove(t)
        tags_removed += len(low)
        if low and arc_failure_code(notes, arc, requirepeak=True) > 0:
            doomed.append(arc)
    for arc in doomed:
        notes['arcs'].remove(arc)
        for t in list(arc.tags):
            arc.tags.remove(t)
            notes['meas'][t.frame].remove(t)
            tags_removed += 1
        say(2, '  removed arc {} starting at frame {}'.format(
            arc.id_, arc.get_frame_range(notes)[0]))
    say(2, '  done: {} detections removed, {} arcs removed'.format(
        tags_removed, len(doomed)))


# ---------------------------------------------------------------------------
#  throws, catches, runs
# ---------------------------------------------------------------------------

def compile_arc_data(notes):
    """Throw/catch frames and positions of every arc relative to the body
    origin, in juggler coordinates (reference tracker.py:1998-2029)."""
    sn, cs = sin(notes['camera_tilt']), cos(notes['camera_tilt'])
    for arc in notes['arcs']:
        ox, oy = notes['origin'][round(arc.f_peak)]
        ox_j = ox * cs - oy * sn
        oy_j = ox * sn + oy * cs
        half_sq = (oy_j - arc.c) / arc.e
        if half_sq > 0:
            half = sqrt(half_sq)
        else:                   # peak below hand level
            first, last = arc.get_tag_range()
            half = max(abs(arc.f_peak - first), abs(arc.f_peak - last))
        arc.f_throw = arc.f_peak - half
        arc.f_catch = arc.f_peak + half
        arc.x_throw = (arc.a - arc.b * half) - ox_j
        arc.x_catch = (arc.a + arc.b * half) - ox_j
        arc.height = oy_j - arc.c


def find_runs(notes):
    """Arcs that overlap in time form a run; runs of one arc are dropped with
    their tags (reference tracker.py:2031-2077)."""
    arcs = notes['arcs']
    if not arcs:
        notes['runs'] = 0
        return []
    runs = []
    horizon = None
    for arc in sorted(arcs, key=lambda a: a.f_throw):
        if runs and arc.f_throw < horizon:
            runs[-1].append(arc)
            horizon = max(horizon, arc.f_catch)
        else:
            runs.append([arc])
            horizon = arc.f_catch
    kept = [run for run in runs if len(run) >= 2]
    loners = [run[0] for run in runs if len(run) < 2]
    notes['arcs'] = [a for a in notes['arcs'] if a not in loners]
    for arc in loners:
        for t in arc.tags:
            notes['meas'][t.frame].remove(t)
    notes['runs'] = len(kept)
    return kept


# ---------------------------------------------------------------------------
#  hands
# ---------------------------------------------------------------------------

def _other(hand):
    return 'right' if hand == 'left' else 'left'


def _assign_side(notes, run, tps, side):
    """Hands for all throws (side='throw') or catches (side='catch') of a
    run (reference tracker.py:2079-2330).  Seeds: the outer quarters in x are
    right / left.  Then hands spread through time: events closer than a
    hand's minimum cycle come from opposite hands, throws that nearly
    coincide in time and place are one multiplex throw, and whatever is left
    alternates from its nearest assigned neighbour (or is split by position
    if it is one of a synchronous pair)."""
    f_of = (lambda a: a.f_throw) if side == 'throw' else (lambda a: a.f_catch)
    x_of = (lambda a: a.x_throw) if side == 'throw' else (lambda a: a.x_catch)
    attr = 'hand_' + side

    def hand(a):
        return getattr(a, attr)

    def give(a, value):
        setattr(a, attr, value)

    by_x = sorted(run, key=x_of)
    n = len(by_x)
    for arc in by_x[:int(n / 4)]:
        give(arc, 'right')
    for arc in by_x[int(3 * n / 4):]:
        give(arc, 'left')

    stack = [a for a in run if hand(a) is not None]
    if not stack:
        seed = max(run, key=lambda a: abs(x_of(a)))
        give(seed, 'right' if x_of(seed) < 0 else 'left')
        stack = [seed]

    fps = notes['fps']
    together_frames = 0.05 * fps
    together_pixels = 10.0 / notes['cm_per_pixel']
    if side == 'throw':
        cycle = 0.23 * fps if tps is None else (1.0 / tps) * 1.3 * fps
    else:
        cycle = 0.18 * fps if tps is None else (1.0 / tps) * 1.0 * fps

    while True:
        while stack:
            known = stack.pop()
            f0, x0 = f_of(known), x_of(known)
            if side == 'throw':
                for arc in [a for a in run if hand(a) is None and
                            f0 - together_frames < f_of(a) <
                            f0 + together_frames and
                            x0 - together_pixels < x_of(a) <
                            x0 + together_pixels]:
                    give(arc, hand(known))
                    stack.append(arc)
            for arc in [a for a in run if hand(a) is None and
                        f0 - cycle < f_of(a) < f0 + cycle]:
                give(arc, _other(hand(known)))
                stack.append(arc)

        pending = [a for a in run if hand(a) is None]
        if not pending:
            return
        settled = [a for a in run if hand(a) is not None]
        nearest = [(a, min(settled, key=lambda b: abs(f_of(a) - f_of(b))))
                   for a in pending]
        arc, neighbour = min(nearest,
                             key=lambda p: abs(f_of(p[0]) - f_of(p[1])))
        twins = [a for a in run if a is not arc and
                 abs(f_of(a) - f_of(arc)) < together_frames]
        if twins:
            twin = twins[0]
            if x_of(arc) > x_of(twin):
                give(arc, 'left')
                give(twin, 'right')
            else:
                give(arc, 'right')
                give(twin, 'left')
            stack.extend((arc, twin))
        else:
            give(arc, _other(hand(neighbour)))
            stack.append(arc)


def assign_hands(notes, run_dict, say=lambda level, text: None):
    run = run_dict['throw']
    say(3, 'Assigning hands to arcs...')
    for arc in run:
        arc.hand_throw = arc.hand_catch = None
    # both sides are seeded before either is propagated, as in the reference;
    # the two passes do not read each other's results
    _assign_side(notes, run, run_dict['throws per sec'], 'throw')
    _assign_side(notes, run, run_dict['throws per sec'], 'catch')
    for arc in run:
        say(3, 'arc {} throwing from {}, catching in {}'.format(
            arc.throw_id, arc.hand_throw, arc.hand_catch))


def connect_arcs(run_dict):
    """Chain the arcs of each ball: an arc continues the most recent earlier
    catch in its throwing hand that has no successor yet (reference
    tracker.py:2332-2363)."""
    run = run_dict['throw']
    for arc in run:
        arc.next = None
    for arc in run:
        arc.prev = max((p for p in run if p.f_catch < arc.f_throw and
                        p.hand_catch == arc.hand_throw and p.next is None),
                       key=lambda p: p.f_catch, default=None)
        if arc.prev is not None:
            arc.prev.next = arc


def estimate_ball_count(notes, run_dict, say=lambda level, text: None):
    """reference tracker.py:2365-2412"""
    run = run_dict['throw']
    tps = run_dict['throws per sec']
    height_cm = notes['cm_per_pixel'] * mean(a.height for a in run)
    if tps is None:
        estimate = rounded = 1
    else:
        dwell_ratio = 0.63      # fraction of the time a hand holds a ball
        estimate = 2 * dwell_ratio + tps * sqrt(8 * height_cm / G_CM_PER_S2)
        same_side = sum(1 for a in run if a.hand_catch == a.hand_throw)
        both_known = sum(1 for a in run if a.hand_catch is not None and
                         a.hand_throw is not None)
        if both_known > 0:
            if same_side > 0.5 * both_known:        # fountain: even count
                rounded = 2 * int(round(0.5 * estimate))
            else:                                   # cascade: odd count
                rounded = 1 + 2 * int(round(0.5 * (estimate - 1)))
        else:
            rounded = int(round(estimate))
    chains = sum(1 for a in run if a.prev is None)
    run_dict['balls'] = min(rounded, chains)
    if tps is not None:
        say(2, 'duration = {:.1f} sec, tps = {:.2f} Hz, height = {:.1f} cm'
            .format(run_dict['duration'], tps, height_cm))
    say(2, 'N_est = {:.2f}, N_round = {}, N_max = {} --> N = {}'.format(
        estimate, rounded, chains, run_dict['balls']))


def analyze_run_form(notes, run_dict, say=lambda level, text: None):
    """Pattern width, target throw/catch points, and per-arc timing errors
    from the third throw on (reference tracker.py:2414-2504)."""
    run = run_dict['throw']
    fps, cm = notes['fps'], notes['cm_per_pixel']
    left = mean(a.x_catch for a in run if a.hand_catch == 'left')
    right = mean(a.x_catch for a in run if a.hand_catch == 'right')
    width = cm * (left - right)
    run_dict['width'] = width
    balls = run_dict['balls']
    scoop = SCOOP_OVER_WIDTH[balls] if balls < 15 else 0.3
    run_dict['target throw point cm'] = (0.5 - scoop) * width
    run_dict['target catch point cm'] = 0.5 * width

    order = sorted(run, key=lambda a: a.throw_id)
    count = len(run)
    for k in range(2, count):
        arc = order[k]
        # throws most likely to collide with this one; the ideal throw time
        # is the straight line through theirs, read off at offset 0
        offsets = (-2, 2) if balls % 2 == 0 else (-2, -1, 1, 2)
        offsets = [d for d in offsets if 0 <= k + d < count]
        if len(offsets) > 2:
            n = sx = sxx = sf = sxf = 0
            for d in offsets:
                f = order[k + d].f_throw
                n += 1
                sx += d
                sxx += d * d
                sf += f
                sxf += d * f
            ideal = (sxx * sf - sx * sxf) / (sxx * n - sx * sx)
            arc.throw_error_s = (arc.f_throw - ideal) / fps
            arc.throw_error_cm = (2.0 * arc.e * fps**2 * cm) * \
                arc.throw_error_s
        # catches: midway between the neighbouring catches of the same hand
        before = max((a for a in run if a.f_catch < arc.f_catch and
                      a.hand_catch == arc.hand_catch),
                     key=lambda a: a.f_catch, default=None)
        after = min((a for a in run if a.f_catch > arc.f_catch and
                     a.hand_catch == arc.hand_catch),
                    key=lambda a: a.f_catch, default=None)
        if before is not None and after is not None:
            ideal = 0.5 * (before.f_catch + after.f_catch)
            arc.catch_error_s = (arc.f_catch - ideal) / fps
            arc.catch_error_cm = (-2.0 * arc.e * fps**2 * cm) * \
                arc.catch_error_s

        def fmt(seconds, cms, what):
            if seconds is None:
                return '{} None'.format(what)
            return '{} {:.3f} s ({:.1f} cm)'.format(what, seconds, cms)
        say(3, 'arc {}: {}, {}'.format(
            arc.throw_id, fmt(arc.throw_error_s, arc.throw_error_cm, 'throw'),
            fmt(arc.catch_error_s, arc.catch_error_cm, 'catch')))


# ---------------------------------------------------------------------------
#  driver
# ---------------------------------------------------------------------------

def analyze_juggling(notes, say=lambda level, text: None, progress=None):
    """reference tracker.py:1785-1865"""
    say(1, 'Juggling analyzer starting...')
    set_body_origin(notes, say)
    remove_tags_below_hands(notes, say)
    compile_arc_data(notes)
    runs = find_runs(notes)
    say(2, 'Number of runs detected = {}'.format(notes['runs']))

    fps = notes['fps']
    notes['run'] = []
    origin_in_runs = dict()
    for run_id, run in enumerate(runs, start=1):
        for throw_id, arc in enumerate(sorted(run, key=lambda a: a.f_throw),
                                       start=1):
            arc.run_id, arc.throw_id = run_id, throw_id
        info = {'throw': run, 'throws': len(run)}
        say(2, '--- Analyzing run {} ------------------------'.format(run_id))
        say(2, 'Number of arcs detected = {}'.format(len(run)))
        first_throw = min(a.f_throw for a in run)
        last_throw = max(a.f_throw for a in run)
        last_catch = max(a.f_catch for a in run)
        info['frame range'] = (first_throw, last_catch)
        info['duration'] = (last_catch - first_throw) / fps
        info['height'] = (notes['cm_per_pixel'] *
                          mean(a.height for a in run[2:])
                          if len(run) > 2 else None)
        info['throws per sec'] = (
            (len(run) - 1) / ((last_throw - first_throw) / fps)
            if last_throw != first_throw else None)

        assign_hands(notes, info, say)
        connect_arcs(info)
        estimate_ball_count(notes, info, say)
        analyze_run_form(notes, info, say)
        notes['run'].append(info)

        for f in range(floor(first_throw), ceil(last_catch) + 1):
            if f in notes['origin']:
                origin_in_runs[f] = notes['origin'][f]
        if progress is not None:
            progress()
    notes['origin'] = origin_in_runs
    say(2, '--------------------------------------------')
    say(1, 'Juggling analyzer done')
