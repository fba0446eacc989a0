#!/opt/conda/bin/python3.9
"""tests/golden/custom.npz: user-written SDFs (tests/fixtures.py CUSTOM_FIXTURES: closures decorated with
the reference's @sdf3 / @op3 / @sdf2, reference README.md:258-295) evaluated and meshed by RUNNING the
unmodified reference:

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden_custom.py

and tests/golden/grid3d.npz: the closure of the reference's `Mesh.sdf` (reference sdf/mesh.py:96-105) on
synthetic voxel grids.  pyopenvdb (the voxeliser, mesh.py:66-94) is not installed, so the closure is
assembled here from the SAME ingredients the reference uses -- its own `box(a=a, b=b)` estimator, scipy's
RegularGridInterpolator(bounds_error=False, fill_value=background) over a float32 array, and the
`np.where(e > background, e, d)` select -- wrapped with the reference's `sdf3` and run through the
reference's `generate`.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def grids():
    """name -> (X, Y, Z, A float32, background, (a, b)): narrow-band level sets like OpenVDB's (values clamped
    to +-background), voxel centres on np.linspace axes; importable under Python 3.10 (the tests rebuild them)"""
    out = {}
    def band(d, bg):
        return np.clip(d, -bg, bg).astype(np.float32)
    vs = 0.08
    ax = [np.linspace(-1.2, 1.2, 31), np.linspace(-1.2, 1.2, 31), np.linspace(-0.56, 0.56, 15)]
    G = np.meshgrid(*ax, indexing='ij')
    torus = np.sqrt((np.sqrt(G[0] ** 2 + G[1] ** 2) - 0.7) ** 2 + G[2] ** 2) - 0.25
    out['torus'] = (ax[0], ax[1], ax[2], band(torus, 3 * vs), 3 * vs, ((-0.95, -0.95, -0.25), (0.95, 0.95, 0.25)))
    vs = 0.05
    ax = [np.linspace(-0.9, 0.85, 36), np.linspace(-0.6, 0.65, 26), np.linspace(-0.75, 0.75, 31)]
    G = np.meshgrid(*ax, indexing='ij')
    two = np.minimum(np.sqrt((G[0] + 0.3) ** 2 + G[1] ** 2 + G[2] ** 2) - 0.45,
                     np.sqrt((G[0] - 0.35) ** 2 + (G[1] - 0.1) ** 2 + (G[2] + 0.2) ** 2) - 0.35)
    out['two_spheres'] = (ax[0], ax[1], ax[2], band(two, 4 * vs), 4 * vs, ((-0.75, -0.45, -0.55), (0.7, 0.45, 0.45)))
    rng = np.random.RandomState(2024)
    ax = [np.linspace(-0.5, 0.5, 9), np.linspace(-0.5, 0.5, 8), np.linspace(-0.5, 0.5, 7)]
    out['noise'] = (ax[0], ax[1], ax[2], rng.uniform(-0.3, 0.3, (9, 8, 7)).astype(np.float32), 0.3,
                    ((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)))
    return out


def grid_points(extent):
    rng = np.random.RandomState(4242)
    p = rng.uniform(-1.4 * extent, 1.4 * extent, (700, 3))
    lat = np.array([(x, y, z) for x in (-extent, 0.0, extent / 2) for y in (-extent / 2, 0.0, extent) for z in (-extent, 0.0, 0.25)])
    far = np.array([(1e9, -1e9, 0.0), (-1e9, 3.0, 2.0), (0.0, 0.0, 1e9), (5.0, 5.0, 5.0)])
    return np.concatenate([p, lat, far])


def main():
    import make_golden as mg                    # imports the reference as `sdf`
    import fixtures
    sdf, core = mg.sdf, mg.core
    P = mg.shared_points()
    out = {'P': P}
    for name in fixtures.CUSTOM_FIXTURES:
        f = fixtures.build(name, mg.NS)
        with np.errstate(all='ignore'):
            out['v_' + name] = f(P.copy()).reshape(-1)
            bounds = core._estimate_bounds(f)
            pts = np.array(core.generate(f, samples=2 ** 17, bounds=bounds, workers=1, verbose=False), dtype=np.float64).reshape(-1, 3)
        out['bounds_' + name] = np.array(bounds, np.float64)
        out['ntri_' + name] = np.int64(len(pts) // 3)
        out['sha_' + name] = np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8)
        out['pts_' + name] = pts
        print(name, len(pts) // 3, 'triangles')
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'custom.npz'), **out)

    from scipy import interpolate
    out = {}
    for name, (X, Y, Z, A, bg, (a, b)) in grids().items():
        estimator = sdf.box(a=a, b=b)
        interpolator = interpolate.RegularGridInterpolator((X, Y, Z), A, bounds_error=False, fill_value=bg)

        @sdf.sdf3
        def mesh_sdf():
            def f(p):                                   # reference sdf/mesh.py:101-105, verbatim semantics
                e = estimator(p)
                d = interpolator(p).reshape((-1, 1))
                return np.where(e > bg, e, d)
            return f
        f = mesh_sdf()
        Pg = grid_points(1.0)
        out['p_' + name] = Pg
        with np.errstate(all='ignore'):
            out['v_' + name] = f(Pg.copy()).reshape(-1)
            g = f.translate((0.05, -0.03, 0.02)) | sdf.sphere(0.2).translate((0, 0, 0.5))
            out['vc_' + name] = g(Pg.copy()).reshape(-1)
            bounds = core._estimate_bounds(g)
            pts = np.array(core.generate(g, samples=2 ** 17, bounds=bounds, workers=1, verbose=False), dtype=np.float64).reshape(-1, 3)
        out['bounds_' + name] = np.array(bounds, np.float64)
        out['ntri_' + name] = np.int64(len(pts) // 3)
        out['sha_' + name] = np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8)
        out['pts_' + name] = pts
        print('grid', name, A.shape, len(pts) // 3, 'triangles')
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'grid3d.npz'), **out)


if __name__ == '__main__':
    main()
