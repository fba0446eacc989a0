#!/opt/conda/bin/python3.9
"""tests/golden/texture.npz: the reference's `image(...)` leaf (reference sdf/text.py:65-153: PIL
conversion, scipy EDT, bilinear texture lookup, fallback rectangle) evaluated by RUNNING the
unmodified reference on synthetic pictures, in 2-D, extruded to 3-D, and meshed end to end.

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden_texture.py

The pictures are made by `pictures()` below from a seeded legacy RandomState, so the tests rebuild
the very same arrays (this file stays importable under Python 3.10 for that).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pictures():
    rng = np.random.RandomState(31337)
    out = {}
    a = np.zeros((40, 60), np.uint8)
    a[10:30, 15:45] = 255
    a[18:22, 25:35] = 0
    out['frame'] = (a, dict(width=3.0))
    yy, xx = np.mgrid[0:96, 0:80]
    b = np.zeros((96, 80), np.uint8)
    for _ in range(9):
        cx, cy, r = rng.uniform(10, 70), rng.uniform(10, 86), rng.uniform(4, 14)
        b[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r] = 255
    out['blobs'] = (b, dict(height=2.0))
    c = (rng.uniform(0, 1, (33, 47)) > 0.55).astype(np.uint8) * 255        # salt and pepper
    out['noise'] = (c, dict(width=1.5, height=1.0))
    return out


def points2(extent):
    rng = np.random.RandomState(99)
    p = rng.uniform(-1.3 * extent, 1.3 * extent, (500, 2))
    lattice = np.array([(x, y) for x in (-extent, -extent / 2, 0.0, extent / 2, extent) for y in (-extent, 0.0, extent / 3, extent)])
    far = np.array([(1e9, -1e9), (-1e9, 3.0), (0.0, 1e9), (5.0, 5.0)])
    return np.concatenate([p, lattice, far])


def main():
    sys.path.insert(0, '/root/reference')
    import sdf
    out = {}
    for name, (arr, kw) in pictures().items():
        f = sdf.image(arr, **kw)
        P = points2(1.6)
        out['p2_' + name] = P
        out['v2_' + name] = f(P).reshape(-1)
        g = f.extrude(0.4)
        P3 = np.concatenate([P, np.linspace(-0.5, 0.5, len(P)).reshape(-1, 1)], axis=1)
        out['v3_' + name] = g(P3).reshape(-1)
        pts = np.array(g.generate(samples=2 ** 15, workers=1, verbose=False), dtype=np.float64).reshape(-1, 3)
        out['gen_ntri_' + name] = np.int64(len(pts) // 3)
        out['gen_sha_' + name] = np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), dtype=np.uint8)
        bounds = sdf.core._estimate_bounds(g)
        out['gen_bounds_' + name] = np.array(bounds, dtype=np.float64)
        print(name, arr.shape, kw, len(pts) // 3, 'triangles')
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'texture.npz'), **out)


if __name__ == '__main__':
    main()
