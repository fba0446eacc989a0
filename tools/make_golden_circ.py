#!/opt/conda/bin/python3.9
"""tests/golden/circ_boundaries.npz: `circular_array` (and twist / bend around it) evaluated by RUNNING the unmodified
reference at the points where a reformulation could part from it -- ON the sector boundaries, a few ulp to 1e-9 rad either
side, the negative x axis with y = +0 / -0, the axis x = y = 0, tiny and huge radii:

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden_circ.py

The device evaluates circular_array in rotation form (csrc/sdf_interp.h L_CIRC_PREP); the children are symmetric about the
x axis, so the reference's value is continuous across a sector boundary and the comparison is meaningful whichever sector a
last-bit difference of arctan2 selects.  `points(count)` and `models(ns, count)` are importable under Python 3.10 (the tests
rebuild the same inputs).

tests/golden/circ_axes.npz (round 5): points ON the coordinate axes (x == 0.0 or y == +-0.0 exactly -- whole planes of a grid
like np.arange(-1, 1, 0.01) -- both signs, many radii) and generic points well inside their sectors, under children that are
NOT symmetric about the x axis: arctan2 is exact on the axes and so is the floored modulo, i.e. the reference puts such a
point into ONE definite sector, and an asymmetric child tells the sectors apart (`axis_points`, `asym_models`)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
COUNTS = (3, 7, 16, 24, 360)


def points(count):
    rng = np.random.RandomState(1000 + count)
    da = 2 * np.pi / count
    ks = np.arange(-count, count + 1)
    ang = (ks * da)[:, None] + np.array([0.0, 1e-16, -1e-16, 3e-13, -3e-13, 1e-9, -1e-9])[None, :]
    ang = np.concatenate([ang.reshape(-1), np.nextafter(ks * da, 10.0), np.nextafter(ks * da, -10.0), rng.uniform(-np.pi, np.pi, 500)])
    r = rng.choice([1e-9, 1e-3, 0.3, 1.0, 1.9, 2.0, 2.1, 7.0, 1e6], size=len(ang))
    P = np.stack([r * np.cos(ang), r * np.sin(ang), rng.uniform(-0.6, 0.6, len(ang))], axis=1)
    extra = np.array([[-1.0, 0.0, 0.1], [-1.0, -0.0, 0.1], [-2.0, 0.0, 0.0], [-2.0, -0.0, 0.0], [0.0, 0.0, 0.2], [-0.0, 0.0, 0.2],
                      [0.0, -0.0, 0.2], [-0.0, -0.0, 0.2], [2.0, 0.0, 0.0], [2.0, -0.0, 0.0], [0.0, 2.0, 0.0], [0.0, -2.0, 0.0],
                      [1e-300, 1e-300, 0.0], [-1e-300, 1e-300, 0.0], [1e150, -1e150, 0.0]])
    return np.ascontiguousarray(np.concatenate([P, extra]))


def models(ns, count):
    return [ns['cylinder'](0.25).circular_array(count, 2),
            ns['sphere'](0.3).circular_array(count, 1.5) | ns['box']((0.2, 0.1, 0.4)).circular_array(count, 0.7),
            ns['rounded_box']((0.6, 0.2, 0.2), 0.05).circular_array(count, 1.0).twist(0.4),
            ns['capsule'](-ns['X'], ns['X'], 0.1).circular_array(count, 0.5).bend(0.3)]


AXIS_COUNTS = (2, 3, 4, 7, 8, 12, 16, 24, 100, 360)


def axis_points(count):
    rng = np.random.RandomState(2000 + count)
    radii = [1e-9, 1e-3, 0.25, 0.3, 0.5, 0.77, 1.0, 1.25, 1.9, 2.0, 2.1, 7.0, 1e6]
    P = []
    for r in radii:
        for z in (0.0, 0.1, -0.23):
            P += [[0.0, r, z], [0.0, -r, z], [-0.0, r, z], [-0.0, -r, z], [r, 0.0, z], [r, -0.0, z], [-r, 0.0, z], [-r, -0.0, z]]
    P = np.array(P)
    # generic points: at least 1e-3 rad away from every sector boundary
    da = 2 * np.pi / count
    k = rng.randint(-count, count, 600)
    ang = (k + rng.uniform(0.001 / da if da > 0.004 else 0.25, 1 - (0.001 / da if da > 0.004 else 0.25), 600)) * da
    r = rng.choice(radii[2:-1], size=600)
    G = np.stack([r * np.cos(ang), r * np.sin(ang), rng.uniform(-0.4, 0.4, 600)], axis=1)
    return np.ascontiguousarray(np.concatenate([P, G]))


def asym_models(ns, count):
    return [ns['box']((0.3, 0.12, 0.2)).translate((0, 0.1, 0)).circular_array(count, 1.0),
            ns['capsule']((-0.1, -0.2, 0), (0.2, 0.15, 0), 0.07).circular_array(count, 0.8) | ns['sphere'](0.2).translate((0.05, 0.3, 0.1)).circular_array(count, 1.9),
            ns['rounded_box']((0.5, 0.1, 0.3), 0.03).rotate(0.5, ns['Z']).circular_array(count, 0.4).translate((0, 0, 0.05))]


def main():
    sys.path.insert(0, '/root/reference')
    import sdf
    ns = {k: getattr(sdf, k) for k in dir(sdf) if not k.startswith('_')}
    out = {}
    for count in COUNTS:
        P = points(count)
        out['P_%d' % count] = P
        for i, f in enumerate(models(ns, count)):
            with np.errstate(all='ignore'):
                out['v_%d_%d' % (count, i)] = np.asarray(f(P), dtype=np.float64).reshape(-1)
    path = os.path.join(ROOT, 'tests', 'golden', 'circ_boundaries.npz')
    np.savez_compressed(path, **out)
    print(path, sum(v.nbytes for v in out.values()), 'bytes of arrays')
    out = {}
    for count in AXIS_COUNTS:
        P = axis_points(count)
        out['P_%d' % count] = P
        for i, f in enumerate(asym_models(ns, count)):
            with np.errstate(all='ignore'):
                out['v_%d_%d' % (count, i)] = np.asarray(f(P), dtype=np.float64).reshape(-1)
    path = os.path.join(ROOT, 'tests', 'golden', 'circ_axes.npz')
    np.savez_compressed(path, **out)
    print(path, sum(v.nbytes for v in out.values()), 'bytes of arrays')


if __name__ == '__main__':
    main()
