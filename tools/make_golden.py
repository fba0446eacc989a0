#!/opt/conda/bin/python3.9
"""Generate tests/golden/*.npz by RUNNING the unmodified reference.

Run in the build container only (it needs /root/reference and the conda interpreter that
has scikit-image 0.18.3):

    env -u PYTHONPATH /opt/conda/bin/python3.9 -W ignore tools/make_golden.py

Nothing here is imported by the product or by the tests; the tests read the .npz files.
Machine dependence: every value comes from correctly rounded NumPy operations EXCEPT the fixtures that go
through BLAS (`np.dot` in rotate / rotate_to / plane / line / cones / polyhedra) or libm -- regenerating on a
different CPU changes those by a few ulp (the tests hold them to `value_tolerance`, not to bit equality).
What is recorded:

  values.npz     reference SDF values  f(P) for every fixture in tests/fixtures.py on one
                 shared point set P (reference sdf/d3.py:24-25 `SDF3.__call__`)
  bounds.npz     reference `_estimate_bounds` (reference sdf/core.py:62-82) per fixture
  mc_*.npz       `sdf.core._marching_cubes` (reference sdf/core.py:16-18 -> skimage 0.18.3
                 `measure.marching_cubes(volume, 0)`) on f64 volumes
  gen_*.npz      `sdf.core.generate` (reference sdf/core.py:84-150) end to end: bounds, step,
                 batch classification counts and the full triangle soup (or its sha256)
  mc_classic_probe.npz  the per-configuration triangle lists skimage emits for the 256 sign
                 configurations of one cell (method='lorensen' and 'lewiner'), from which
                 tools/derive_mc_tables.py builds the lookup table used by oracle and kernels
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import sdf  # the reference  # noqa: E402
from sdf import core  # noqa: E402
from skimage import measure  # noqa: E402

import fixtures  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
NS = {k: getattr(sdf, k) for k in dir(sdf) if not k.startswith('_')}


def shared_points():
    rng = np.random.RandomState(12345)
    a = rng.standard_normal((400, 3)) * 1.5
    b = rng.uniform(-4, 4, (100, 3))
    c = rng.randint(-12, 13, (60, 3)) * 0.25          # lattice: hits axes / planes exactly
    d = np.array([
        (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1),
        (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1), (-1, -1, -1), (0.5, 0.5, 0.5),
        (2, 0, 0), (0, 2, 0), (0, 0, 2), (1e-9, 0, 0), (0, 1e-9, 1e-9), (3, 3, 3),
        (0.75, 0, 0), (0, 0.75, 0), (0, 0, 0.75), (0.5, 0, 0), (0, 0.5, 0), (0, 0, 0.5),
        (10, -7, 3), (-25, 14, 2), (1e3, 1e3, -1e3), (1e9, -1e9, 1e9), (-1e9, 1e9, 0.0),
        (0.25, 0.25, 0), (1.5, 1.5, 0), (0, -1.5, 0.25), (-0.5, 0.5, -0.5), (2.7, 5.4, 0),
        (1.35, 2.7, 0), (1.35, 0, 0.1), (0, 2.7, -0.1), (12, 0, 0), (0, 10, 0.5),
    ], dtype=float)
    return np.concatenate([a, b, c, d]).astype(np.float64)


def gen_values():
    P = shared_points()
    out = {'P': P}
    for name in fixtures.FIXTURES:
        f = fixtures.build(name, NS)
        with np.errstate(all='ignore'):
            v = f(P.copy()).reshape(-1)
        assert v.dtype == np.float64 and v.shape == (len(P),)
        out['v_' + name] = v
    np.savez_compressed(os.path.join(OUT, 'values.npz'), **out)
    print('values.npz', len(fixtures.FIXTURES), 'fixtures x', len(P), 'points')


def gen_bounds():
    out = {}
    for name in fixtures.FIXTURES:
        if name in ('ex_custbox',):      # 1e9-sized model: estimator output not meaningful
            continue
        f = fixtures.build(name, NS)
        try:
            with np.errstate(all='ignore'):
                b = core._estimate_bounds(f)
            out[name] = np.array(b, dtype=np.float64)
        except Exception as e:           # reference crashes on empty `where` (core.py:63 TODO)
            print('bounds failed for', name, type(e).__name__)
    np.savez_compressed(os.path.join(OUT, 'bounds.npz'), **out)
    print('bounds.npz', len(out))


def mc_soup(volume):
    try:
        return core._marching_cubes(volume).astype(np.float32)
    except Exception:
        return np.zeros((0, 3), np.float32)


def gen_mc():
    rng = np.random.RandomState(777)
    vols = {}
    # smooth fields sampled like a batch (33^3), several shifts
    g = np.arange(33, dtype=np.float64)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    vols['sphere33'] = np.sqrt((X - 15.3) ** 2 + (Y - 16.1) ** 2 + (Z - 14.7) ** 2) - 11.37
    vols['torus33'] = np.sqrt((np.sqrt((X - 16.2) ** 2 + (Y - 15.9) ** 2) - 9.1) ** 2 + (Z - 16.4) ** 2) - 3.3
    vols['gyroid33'] = (np.sin(X * 0.41) * np.cos(Y * 0.37) + np.sin(Y * 0.43) * np.cos(Z * 0.39)
                        + np.sin(Z * 0.4) * np.cos(X * 0.38)) * 3.0 + 0.21
    vols['plane_grid_aligned'] = X - 7.0                       # exact zeros on a lattice plane
    vols['plane_offset'] = 0.3 * X + 0.5 * Y - 0.2 * Z - 7.77
    vols['two_spheres'] = np.minimum(np.sqrt((X - 9) ** 2 + (Y - 9) ** 2 + (Z - 9) ** 2) - 6.5,
                                     np.sqrt((X - 22) ** 2 + (Y - 22) ** 2 + (Z - 22) ** 2) - 6.5)
    vols['ragged_2x5x9'] = rng.standard_normal((2, 5, 9))
    vols['ragged_33x2x17'] = rng.standard_normal((33, 2, 17))
    vols['cell_2x2x2'] = rng.standard_normal((2, 2, 2))
    vols['noise12'] = rng.standard_normal((12, 12, 12))       # every configuration, ambiguous ones too
    vols['noise20_sparse'] = rng.standard_normal((20, 20, 20)) + 1.2
    vols['all_positive'] = np.abs(rng.standard_normal((6, 6, 6))) + 0.1   # ValueError -> empty
    vols['all_negative'] = -np.abs(rng.standard_normal((6, 6, 6))) - 0.1
    vols['single_sample_axis'] = rng.standard_normal((1, 8, 8))           # ValueError -> empty
    z = rng.standard_normal((9, 9, 9)); z[rng.uniform(size=z.shape) < 0.15] = 0.0
    vols['noise_with_zeros'] = z
    tiny = rng.standard_normal((8, 8, 8)) * 1e-30                          # f32 cast -> denormal/zero
    vols['tiny_values'] = tiny
    out = {}
    for k, v in vols.items():
        out['vol_' + k] = v.astype(np.float64)
        out['soup_' + k] = mc_soup(v)
        print('  mc', k, v.shape, len(out['soup_' + k]) // 3, 'tris')
    np.savez_compressed(os.path.join(OUT, 'mc_volumes.npz'), **out)


def gen_mc_probe():
    """Triangle lists of one cell for all 256 sign configurations.

    Corner c of the 2x2x2 volume is (o0,o1,o2) with c = 4*o0 + 2*o1 + o2 (volume axis
    order); bit c of the configuration is set when the sample is > 0.  Magnitudes differ per
    corner so that every crossing position identifies its edge.  Edge key = (axis, oa, ob)
    with (oa, ob) the offsets on the two other axes in increasing axis order."""
    rng = np.random.RandomState(99)

    def edges(s):
        out = []
        for p in s:
            fr = [i for i in range(3) if p[i] not in (0.0, 1.0)]
            if len(fr) != 1:
                return None            # a centre vertex (Lewiner tilings with a 13th vertex)
            a = fr[0]
            o = [int(p[i]) for i in range(3) if i != a]
            out.append(a * 4 + o[0] * 2 + o[1])
        return out

    classic = -np.ones((256, 15), np.int8)
    lewiner_same = np.zeros(256, np.uint8)
    for cfg in range(256):
        seen_c, seen_l = set(), set()
        for trial in range(64):
            mag = rng.uniform(0.1, 1.0, 8)
            vol = np.array([mag[c] if (cfg >> c) & 1 else -mag[c] for c in range(8)]).reshape(2, 2, 2)
            for method, seen in (('lorensen', seen_c), ('lewiner', seen_l)):
                try:
                    v, f, _, _ = measure.marching_cubes(vol, 0, method=method)
                    e = edges(v[f].reshape(-1, 3))
                except (ValueError, RuntimeError):
                    e = []
                seen.add(tuple(e) if e is not None else ('centre',))
        assert len(seen_c) == 1, cfg
        e = list(seen_c)[0]
        classic[cfg, :len(e)] = e
        lewiner_same[cfg] = 1 if seen_l == seen_c else 0
    np.savez_compressed(os.path.join(OUT, 'mc_classic_probe.npz'),
                        classic=classic, lewiner_same=lewiner_same)
    print('mc_classic_probe.npz: value-independent & identical under lewiner:', int(lewiner_same.sum()))


class Counter:
    """wraps core._worker to record the per-batch classification in batch order"""
    def __init__(self):
        self.kinds = []
        self.orig = core._worker

    def __call__(self, sdf_, job, step, sparse):
        r = self.orig(sdf_, job, step, sparse)
        self.kinds.append(0 if r is None else (1 if len(r) == 0 else 2))
        return r


def run_generate(name, full, **kw):
    f = fixtures.build(name, NS)
    with np.errstate(all='ignore'):
        bounds = kw.pop('bounds', None) or core._estimate_bounds(f)
    (x0, y0, z0), (x1, y1, z1) = bounds
    samples = kw.pop('samples', 2 ** 22)
    step = kw.pop('step', None)
    if step is None:
        step = ((x1 - x0) * (y1 - y0) * (z1 - z0) / samples) ** (1 / 3)
    cnt = Counter()
    core._worker = cnt
    t0 = time.time()
    try:
        with np.errstate(all='ignore'):
            pts = core.generate(f, step=step, bounds=bounds, workers=1, verbose=False, **kw)
    finally:
        core._worker = cnt.orig
    dt = time.time() - t0
    pts = np.array(pts, dtype=np.float64).reshape(-1, 3)
    kinds = np.array(cnt.kinds, np.uint8)
    rec = {
        'bounds': np.array(bounds, np.float64),
        'step': np.array(step, np.float64),
        'kinds': kinds,                       # 0 skipped / 1 empty / 2 nonempty, batch order
        'ntri': np.array(len(pts) // 3),
        'sha256': np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8),
        'seconds': np.array(dt),
    }
    if full:
        rec['points'] = pts
    print('  gen %-12s tris %8d  batches %5d (s/e/n %d/%d/%d)  %.2fs' % (
        name, len(pts) // 3, len(kinds), (kinds == 0).sum(), (kinds == 1).sum(), (kinds == 2).sum(), dt))
    return rec


def gen_generate():
    jobs = [
        # tag, fixture, full soup?, kwargs
        ('example_s15', 'ex_example', True, dict(samples=2 ** 15)),
        ('example_s17', 'ex_example', True, dict(samples=2 ** 17)),
        ('example_s17_dense', 'ex_example', False, dict(samples=2 ** 17, sparse=False)),
        ('example_s17_b8', 'ex_example', False, dict(samples=2 ** 17, batch_size=8)),
        ('example_s17_aniso', 'ex_example', False, dict(step=(0.031, 0.043, 0.037))),
        ('example_s22', 'ex_example', False, dict(samples=2 ** 22)),          # BASELINE config 1
        ('gearlike_s16', 'ex_gearlike', True, dict(samples=2 ** 16)),
        ('gearlike_s20', 'ex_gearlike', False, dict(samples=2 ** 20)),
        ('blobby_s16', 'ex_blobby', True, dict(samples=2 ** 16)),
        ('blobby_s20', 'ex_blobby', False, dict(samples=2 ** 20)),
        ('weave_s16', 'ex_weave', True, dict(samples=2 ** 16)),
        ('weave_s19', 'ex_weave', False, dict(samples=2 ** 19)),
        ('knurling_s16', 'ex_knurling', True, dict(samples=2 ** 16)),
        ('pawn_s16', 'ex_pawn', True, dict(samples=2 ** 16)),
        ('bend_radial_s16', 'bend_radial', False, dict(samples=2 ** 16, sparse=False)),
        ('torus_s15', 'torus', True, dict(samples=2 ** 15)),
        ('slice_s15', 'slice', False, dict(samples=2 ** 15)),
        ('extrude_to_s15', 'extrude_to', False, dict(samples=2 ** 15)),
        # batch_size > 32 (reference sdf/core.py:87, 114-119 takes any): round 5
        ('example_s17_b64', 'ex_example', True, dict(samples=2 ** 17, batch_size=64)),
        ('example_s22_b48', 'ex_example', False, dict(samples=2 ** 22, batch_size=48)),
        ('example_s22_b128', 'ex_example', False, dict(samples=2 ** 22, batch_size=128)),
        ('gearlike_s20_b64', 'ex_gearlike', False, dict(samples=2 ** 20, batch_size=64)),
        ('blobby_s20_b40_dense', 'ex_blobby', False, dict(samples=2 ** 20, batch_size=40, sparse=False)),
    ]
    only = os.environ.get('GOLDEN_ONLY')       # (comma-separated tags: make just those)
    for tag, name, full, kw in jobs:
        if only and tag not in only.split(','):
            continue
        rec = run_generate(name, full, **dict(kw))
        rec['fixture'] = np.array(name)
        rec['kwargs'] = np.array(repr(kw))
        np.savez_compressed(os.path.join(OUT, 'gen_%s.npz' % tag), **rec)


def gen_stl():
    """byte-exact STL of a small soup (reference sdf/stl.py:4-24)"""
    import tempfile
    d = np.load(os.path.join(OUT, 'gen_example_s15.npz'))
    pts = d['points']
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, 'a.stl')
        sdf.write_binary_stl(p, list(pts))
        raw = open(p, 'rb').read()
    np.savez_compressed(os.path.join(OUT, 'stl_example_s15.npz'),
                        stl=np.frombuffer(raw, np.uint8))
    print('stl bytes', len(raw))


if __name__ == '__main__':
    what = sys.argv[1:] or ['values', 'bounds', 'mc', 'probe', 'generate', 'stl']
    if 'values' in what: gen_values()
    if 'bounds' in what: gen_bounds()
    if 'mc' in what: gen_mc()
    if 'probe' in what: gen_mc_probe()
    if 'generate' in what: gen_generate()
    if 'stl' in what: gen_stl()
