#!/opt/conda/bin/python3.9
"""tests/golden/mc33_volumes.npz: volumes that exercise every Lewiner case (ambiguous faces,
interior tests, centre vertices) with the soups skimage 0.18.3 returns for them through the
reference's own call (reference sdf/core.py:16-18: measure.marching_cubes(volume, 0), verts[faces]).

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden_mc33.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
from sdf import core  # noqa: E402
from scipy import ndimage  # noqa: E402


def main():
    rng = np.random.RandomState(2024)
    vols = {}
    for i, n in enumerate((9, 14, 14)):
        vols['noise%d' % i] = rng.standard_normal((n, n + 1, n + 2))
    vols['smooth'] = ndimage.gaussian_filter(rng.standard_normal((26, 24, 22)), 1.2)
    vols['ints'] = rng.randint(-2, 3, (10, 11, 9)).astype(float)          # exact zeros and ties
    vols['halves'] = rng.randint(-3, 4, (9, 9, 9)) * 0.5 + 0.25
    g = np.mgrid[-1:1:15j, -1:1:16j, -1:1:17j]
    vols['saddle'] = g[0] * g[1] - 0.3 * g[2] + 0.01                      # bilinear saddles on every face
    vols['two_spheres'] = np.minimum(np.linalg.norm(g - 0.33, axis=0), np.linalg.norm(g + 0.33, axis=0)) - 0.55
    out = {}
    for name, v in vols.items():
        v = np.ascontiguousarray(v, dtype=np.float64)
        soup = core._marching_cubes(v)
        out['vol_' + name] = v.astype(np.float32)
        out['soup_' + name] = np.ascontiguousarray(soup, dtype=np.float32)
        print(name, v.shape, len(soup) // 3, 'triangles')
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'mc33_volumes.npz'), **out)


if __name__ == '__main__':
    main()
