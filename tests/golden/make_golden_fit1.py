#!/usr/bin/env python3
"""
Golden vectors for the M-step's skip paths, from the REFERENCE's own
fit_arcs() (reference tracker.py:698-798) with unit weights -- the form the
native library can be driven in through hk_clip_set_arcs + hk_clip_fit_arcs
(tags without a `weight` attribute count 1.0, tracker.py:722-725).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_fit1.py

Writes fit_unit_vectors.json next to itself; unit_vectors.json and the
run_*.json files are not touched (the reference sums in set order, so
regenerating them would move their last bits).
"""
import json
import os

import numpy as np

import make_golden as G         # binds `hawkeye` to /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))


def tags_on_parabola(rng, frames, fp, e, noise=0.5):
    return [[int(f), 400.0 + 2.5 * (f - fp) + rng.normal(0, noise),
             150.0 + e * (f - fp)**2 + rng.normal(0, noise)] for f in frames]


def cases():
    rng = np.random.RandomState(20261111)
    pick = lambda fp, n: sorted(rng.choice(                 # noqa: E731
        np.arange(int(fp) - 25, int(fp) + 25), n, replace=False).tolist())
    out = []
    out.append(('fit_12_tags', 0.0, 51.7,
                tags_on_parabola(rng, pick(50.0, 12), 50.0, 0.45)))
    out.append(('fit_30_tags_tilted', 0.03, 55.0,
                tags_on_parabola(rng, pick(53.3, 30), 53.3, 0.55)))
    out.append(('fit_exactly_3_tags', 0.0, 58.3,
                tags_on_parabola(rng, [51, 54, 55], 56.6, 0.65)))
    out.append(('skip_fewer_than_3_tags', 0.0, 61.6,
                tags_on_parabola(rng, [36, 60], 59.9, 0.75)))
    same = tags_on_parabola(rng, [63] * 8, 63.2, 0.85, noise=40.0)
    out.append(('skip_singular_all_tags_in_one_frame', -0.02, 64.9, same))
    two = tags_on_parabola(rng, [70, 70, 70, 71, 71, 71], 66.5, 0.95)
    out.append(('skip_singular_two_distinct_frames', 0.0, 68.2, two))
    flat = [[f, 400.0 + 2.5 * f, 0.0] for f in (40, 43, 44, 47, 51, 52)]
    out.append(('skip_e_equal_zero', 0.0, 45.5, flat))
    nan = tags_on_parabola(rng, pick(80.0, 9), 80.0, 0.5)
    nan[4][1] = float('nan')
    out.append(('skip_nan', 0.0, 81.0, nan))
    return out


def main():
    fits = []
    for name, tilt, f_peak, tags in cases():
        notes = G.step2_notes({}, 1280, 720, 30.0, 200)
        notes['camera_tilt'] = tilt
        sc = G.make_scanner(notes)
        arc = G.ref_types.Ballarc()
        arc.f_peak, arc.a, arc.b, arc.c, arc.e = f_peak, 1.0, 2.0, 3.0, 0.4
        start = [arc.a, arc.b, arc.c, arc.e, arc.f_peak]
        for (f, x, y) in tags:
            arc.tags.add(G.ref_types.Balltag(f, x, y, 16.0))   # no .weight
        sc.fit_arcs([arc])
        result = [arc.a, arc.b, arc.c, arc.e, arc.f_peak]
        fits.append({'name': name, 'tilt': tilt, 'start': start, 'tags': tags,
                     'result': result, 'skipped': result == start})
        print(name, 'skipped' if result == start else result)
    with open(os.path.join(HERE, 'fit_unit_vectors.json'), 'w') as fh:
        json.dump({'fit_arcs_unit_weights': fits}, fh, indent=1)


if __name__ == '__main__':
    main()
