#!/usr/bin/env python3
"""Randomised steps 3-4 parity run: random tag sets (geometry, fps, ball
count, clutter, missed detections, tilt, slow distractor tracks) through the
native code behind the C ABI, against the Python oracle (oracle/em_ref):
same arcs with the same member lists, parameters to 1e-6 relative, same
surviving tags and hard assignments.  Step 3 is host code (runs without a
GPU); --step4 needs the GPU for the EM kernels.

    python tests/fuzz_arcs.py [--seconds 120] [--seed 1] [--step4]

The parabola fit of the reference is ill-conditioned while a trial arc's peak
frame lies far from its tags (tracker.py:757 sums t**4 with t in the
hundreds), so now and then two correct implementations that round
differently grow a trial arc differently and end with different arc sets --
the reference does that against itself, as it sums in set order.  Where
/root/reference is present (the build container) a mismatch is therefore
arbitrated by running the reference's own build_initial_arcs() on the case
(step 3 only): the case counts against the native code only if the oracle
agrees with the reference and the native code does not.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd import synth                               # noqa: E402
from hawkeye_amd.arcs import Clip                           # noqa: E402
from oracle import em_ref                                   # noqa: E402

REL = 1e-6
em_ref.POWERS_AS_PRODUCTS = True    # round like the native code (see there)


def close(a, b, rel=REL, abs_=1e-9):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


def random_case(rng):
    w, h = [(640, 480), (1280, 720), (1920, 1080), (960, 540), (320, 240),
            (853, 481)][rng.randint(6)]
    fps = float(rng.choice([24, 30, 50, 60, 120]))
    frames = int(rng.randint(40, 220))
    balls = int(rng.randint(1, 10))
    scene = synth.JugglingScene(w, h, fps, balls, frames,
                                int(rng.randint(1 << 30)),
                                tilt_deg=float(rng.choice([0, 0, 2.5, -4])))
    meas = synth.make_tags(scene, int(rng.randint(1 << 30)),
                           clutter_per_frame=float(rng.choice([0, 2, 6, 14])),
                           detect_prob=float(rng.choice([1.0, 0.92, 0.7])),
                           pos_sigma=float(rng.choice([0.3, 0.6, 2.0])))
    for d in range(int(rng.choice([0, 0, 3]))):     # slow distractor tracks
        px, py = rng.uniform(0, w), rng.uniform(0, h)
        for f in range(frames):
            px += rng.normal(0, 1.0)
            py += rng.normal(0, 1.0)
            meas.setdefault(f, []).append((float(px), float(py),
                                           float(h / 45.0 + d)))
    return dict(width=w, height=h, fps=fps, frames=frames, balls=balls), meas


def compare(spec, meas, step4):
    w, h, fps, frames = (spec['width'], spec['height'], spec['fps'],
                         spec['frames'])
    frame, x, y, r, tw, start = synth.tags_to_soa(meas, frames)
    kw = {} if step4 else {'stream': 0}
    clip = Clip(w, h, fps, frames, frame, x, y, r, tw, **kw)
    sc = em_ref.Scene(meas, w, h, fps, frame_count=frames)
    try:
        clip.build_initial_arcs()
        sc.build_initial_arcs()
        err = compare_arcs(clip, sc, 'step 3')
        if err or not step4:
            return err
        if sc.g is not None:
            st = clip.state()
            if not close(st['g_px_per_frame_sq'], sc.g):
                return 'step 3: g differs'
        clip.em_optimize()
        sc.em_optimize()
        err = compare_arcs(clip, sc, 'step 4')
        if err:
            return err
        if not close(clip.state()['camera_tilt'], sc.tilt, abs_=1e-12):
            return 'step 4: camera tilt {} vs {}'.format(
                clip.state()['camera_tilt'], sc.tilt)
        alive, final = clip.tag_state()
        if list(alive) != list(sc.alive):
            return 'step 4: surviving tags differ'
        # hard assignment: same tags on every arc (arcs matched by members)
        params, ids, tags = clip.arcs()
        mine = sorted(tuple(int(i) for i in range(clip.n)
                            if alive[i] and final[i] == k)
                      for k in range(len(params)))
        theirs = sorted(tuple(i for i in range(sc.n)
                              if sc.alive[i] and sc.final_arc[i] is a)
                        for a in sc.arcs)
        if mine != theirs:
            return 'step 4: hard assignments differ'
        return None
    finally:
        clip.close()


def compare_arcs(clip, sc, what):
    params, ids, tags = clip.arcs()
    if len(params) != len(sc.arcs):
        return '{}: {} arcs, oracle {}'.format(what, len(params), len(sc.arcs))
    # arcs are listed by f_peak; two arcs on the same ideal parabola (noise-
    # free input) have peaks equal to rounding and may come in either order
    mine = sorted(zip(params, tags),
                  key=lambda pt: (round(float(pt[0][4]), 6), int(pt[1][0])))
    theirs = sorted(sc.arcs,
                    key=lambda a: (round(a.params()[4], 6), a.tags[0]))
    for k, ((p, t), arc) in enumerate(zip(mine, theirs)):
        if list(t) != list(arc.tags):
            return '{}: arc {} members differ'.format(what, k)
        for got, want in zip(p, arc.params()):
            if not close(got, want):
                return '{}: arc {} parameter {} vs {}'.format(what, k, got,
                                                              want)
    return None


def reference_step3(spec, meas):
    """[(f_peak, member count)] of the reference's own step 3, or None where
    the reference is not available."""
    if not os.path.isdir('/root/reference/hawkeye'):
        return None
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(
        __file__)), 'golden'))
    import make_golden as mg
    notes = mg.step2_notes(meas, spec['width'], spec['height'], spec['fps'],
                           spec['frames'])
    sc = mg.make_scanner(notes)
    sc.build_initial_arcs()
    return sorted((round(a.f_peak, 4), len(a.tags)) for a in notes['arcs'])


def arbitrate(spec, meas, runs=4):
    """Who agrees with the reference on the case's step-3 arcs: 'native',
    'oracle', 'both' or 'neither' (None without the reference), and how many
    different answers the reference itself gave in `runs` runs (it sums in
    set order, which changes from run to run)."""
    want = [reference_step3(spec, meas) for _ in range(runs)]
    if want[0] is None:
        return None, 0
    frames = spec['frames']
    frame, x, y, r, tw, start = synth.tags_to_soa(meas, frames)
    clip = Clip(spec['width'], spec['height'], spec['fps'], frames, frame, x,
                y, r, tw, stream=0)
    clip.build_initial_arcs()
    params, ids, tags = clip.arcs()
    nat_arcs = sorted((round(float(p[4]), 4), len(t))
                      for p, t in zip(params, tags))
    clip.close()
    sc = em_ref.Scene(meas, spec['width'], spec['height'], spec['fps'],
                      frame_count=frames)
    sc.build_initial_arcs()
    orc = sorted((round(a.params()[4], 4), len(a.tags)) for a in sc.arcs)
    n, o = nat_arcs in want, orc in want
    distinct = len({tuple(w) for w in want})
    return ('both' if n and o else 'native' if n else 'oracle' if o
            else 'neither'), distinct


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=120.0)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--step4', action='store_true')
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    t0 = said = time.time()
    cases = 0
    tally = {}
    while time.time() - t0 < args.seconds:
        state = rng.get_state()[1][:4].tolist()
        spec, meas = random_case(rng)
        err = compare(spec, meas, args.step4)
        cases += 1
        if time.time() - said > 30:     # a silent GPU run is taken to be hung
            print('...', cases, 'cases', flush=True)
            said = time.time()
        if err:
            who, distinct = arbitrate(spec, meas)
            print('MISMATCH', err, spec, 'case', cases, 'rng', state,
                  '-> agrees with the reference:', who,
                  '(the reference gave {} different answers in 4 runs)'.format(
                      distinct), flush=True)
            tally[who] = tally.get(who, 0) + 1
            if who == 'oracle' and distinct == 1:
                sys.exit(1)         # the native code alone disagrees
        if cases % 5 == 0:
            print('{} cases, {:.0f} s'.format(cases, time.time() - t0),
                  flush=True)
    print('ok: {} cases; mismatches arbitrated by the reference: {}'.format(
        cases, tally))


if __name__ == '__main__':
    main()
