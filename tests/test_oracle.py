"""The CPU checker (oracle/) against the golden vectors produced by RUNNING the unmodified
reference (tools/make_golden.py).  This is what pins the oracle; everything else compares
the HIP path with the oracle."""
import glob
import hashlib
import os

import numpy as np
import pytest

import fixtures
from conftest import GOLDEN, ROOT, value_tolerance
from sdf_amd import core


@pytest.mark.parametrize('name', sorted(fixtures.FIXTURES))
def test_values_match_reference(name, ns, golden_values, oracle_lib):
    P = golden_values['P']
    ref = golden_values['v_' + name]
    f = fixtures.build(name, ns)
    v = oracle_lib.evaluate(f, P)
    assert np.array_equal(np.isnan(v), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.all(np.abs(v[ok] - ref[ok]) <= value_tolerance(ref[ok], P[ok]))


def test_values_bit_exact_where_numpy_is_deterministic(ns, golden_values, oracle_lib):
    """fixtures that use only correctly-rounded elementwise NumPy ops must agree bit for bit.  (`rotate` by a
    general angle is NOT among them: the reference's np.dot goes through BLAS, whose kernel choice -- FMA or not --
    depends on the host CPU; goldens regenerated on another machine differ from these by <= 4 ulp there, and
    the tolerance test above covers it.  `orient` and the rotations inside ex_example are by axis-aligned
    matrices of 0 / +-1, exact under either kernel.)"""
    P = golden_values['P']
    for name in ('ex_example', 'ex_blobby', 'ex_pawn', 'sphere', 'box2', 'rounded_box', 'wireframe_box',
                 'torus', 'capsule', 'capped_cylinder', 'rounded_cylinder', 'capped_cone', 'ellipsoid',
                 'pyramid', 'tetrahedron', 'smooth_union', 'smooth_difference', 'smooth_intersection',
                 'blend_k', 'elongate', 'orient', 'scale', 'slab_k', 'rectangle',
                 'rounded_rectangle', 'equilateral_triangle', 'rounded_x', 'vesica', 'slice', 'extrude_to',
                 'ease_in_out_quad', 'ease_out_bounce', 'ease_in_out_back', 'ease_in_out_circ'):
        v = oracle_lib.evaluate(fixtures.build(name, ns), P)
        ref = golden_values['v_' + name]
        assert np.array_equal(v, ref, equal_nan=True), name


def test_bounds_match_reference(ns, oracle_lib):
    b = np.load(os.path.join(GOLDEN, 'bounds.npz'))
    for name in b.files:
        got = np.array(oracle_lib.estimate_bounds(fixtures.build(name, ns)))
        assert np.array_equal(got, b[name]), name


MC = np.load(os.path.join(GOLDEN, 'mc_volumes.npz'))
MC_NAMES = sorted(k[4:] for k in MC.files if k.startswith('vol_'))


@pytest.mark.parametrize('name', MC_NAMES)
def test_marching_cubes_matches_skimage(name, oracle_lib):
    soup, namb = oracle_lib.marching_cubes(MC['vol_' + name])
    ref = MC['soup_' + name]
    # bit-identical soup in skimage's order, ambiguous (Lewiner-tested) cells included
    assert soup.shape == ref.shape
    assert np.array_equal(soup.view(np.uint32), ref.view(np.uint32))


MC33 = np.load(os.path.join(GOLDEN, 'mc33_volumes.npz'))
MC33_NAMES = sorted(k[4:] for k in MC33.files if k.startswith('vol_'))


@pytest.mark.parametrize('name', MC33_NAMES)
def test_marching_cubes_lewiner_cases_match_skimage(name, oracle_lib):
    """noise / integer / saddle volumes that reach every Lewiner case, subcase and the centre
    vertex (tools/make_golden_mc33.py): the soup must be skimage's bit for bit"""
    soup, namb = oracle_lib.marching_cubes(MC33['vol_' + name])
    ref = MC33['soup_' + name]
    assert soup.shape == ref.shape
    assert np.array_equal(soup.view(np.uint32), ref.view(np.uint32))
    if name.startswith('noise'):
        assert namb > 100


GEN = sorted(glob.glob(os.path.join(GOLDEN, 'gen_*.npz')))


@pytest.mark.parametrize('path', GEN, ids=[os.path.basename(p)[4:-4] for p in GEN])
def test_generate_matches_reference(path, ns, oracle_lib):
    d = np.load(path)
    kw = eval(str(d['kwargs']))
    f = fixtures.build(str(d['fixture']), ns)
    bounds = tuple(map(tuple, d['bounds']))
    X, Y, Z, _ = core.grid_axes(bounds, d['step'].tolist())
    r = oracle_lib.generate(f, X, Y, Z, kw.get('batch_size', 32), kw.get('sparse', True))
    assert np.array_equal(r.kinds, d['kinds'])          # skipped / empty / nonempty per batch
    assert len(r.points) // 3 == int(d['ntri'])
    assert hashlib.sha256(r.points.tobytes()).digest() == d['sha256'].tobytes()
    if 'points' in d.files:
        assert np.array_equal(r.points, d['points'])


# ---- sampled 2-D field leaves (reference sdf/text.py) ------------------------------------------
def _pictures():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_texture', os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), 'tools', 'make_golden_texture.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.pictures()


TEX = np.load(os.path.join(GOLDEN, 'texture.npz'))


@pytest.mark.parametrize('name', ['frame', 'blobs', 'noise'])
def test_image_leaf_matches_reference(name, ns, oracle_lib):
    """`image(array)`: PIL conversion + EDT on the host, bilinear lookup + fallback rectangle in the
    checker: the reference's values bit for bit, in 2-D, extruded, and meshed end to end"""
    arr, kw = _pictures()[name]
    f = ns['image'](arr, **kw)
    P = TEX['p2_' + name]
    assert np.array_equal(oracle_lib.evaluate(f, P), TEX['v2_' + name], equal_nan=True)
    g = f.extrude(0.4)
    P3 = np.concatenate([P, np.linspace(-0.5, 0.5, len(P)).reshape(-1, 1)], axis=1)
    assert np.array_equal(oracle_lib.evaluate(g, P3), TEX['v3_' + name], equal_nan=True)
    bounds = tuple(map(tuple, TEX['gen_bounds_' + name]))
    assert np.array_equal(np.array(oracle_lib.estimate_bounds(g)), TEX['gen_bounds_' + name])
    X, Y, Z, _ = core.grid_axes(bounds, samples=2 ** 15)
    r = oracle_lib.generate(g, X, Y, Z, 32, True)
    assert len(r.points) // 3 == int(TEX['gen_ntri_' + name])
    assert hashlib.sha256(r.points.tobytes()).digest() == TEX['gen_sha_' + name].tobytes()


def test_oracle_matches_reference_on_circular_array_boundaries(ns, oracle_lib):
    """tests/golden/circ_boundaries.npz (tools/make_golden_circ.py: the unmodified reference on points ON the sector
    boundaries of circular_array, ulps to 1e-9 rad beside them, the negative x axis with +-0, the axis, tiny and huge
    radii; children symmetric about the x axis, with twist / bend around): the checker agrees to the tolerance of the
    libm models (NumPy's sin / cos / arctan2 vs glibc's)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import make_golden_circ as mgc
    from conftest import value_tolerance
    g = np.load(os.path.join(GOLDEN, 'circ_boundaries.npz'))
    for count in mgc.COUNTS:
        P = g['P_%d' % count]
        for i, f in enumerate(mgc.models(ns, count)):
            ref = g['v_%d_%d' % (count, i)]
            o = oracle_lib.evaluate(f, P)
            ok = np.isfinite(ref)
            assert np.array_equal(np.isfinite(o), ok)
            assert np.all(np.abs(o[ok] - ref[ok]) <= value_tolerance(ref[ok], P[ok])), (count, i)


def test_oracle_matches_reference_on_axis_points_under_asymmetric_children(ns, oracle_lib):
    """tests/golden/circ_axes.npz (tools/make_golden_circ.py, round 5): points ON the coordinate axes -- where arctan2 and
    the floored modulo are exact, so the reference (sdf/d3.py:379-392) puts each into one definite sector -- and generic
    points, under children that are NOT symmetric about the x axis, which tell neighbouring sectors apart"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import make_golden_circ as mgc
    from conftest import value_tolerance
    g = np.load(os.path.join(GOLDEN, 'circ_axes.npz'))
    for count in mgc.AXIS_COUNTS:
        P = g['P_%d' % count]
        assert np.array_equal(P, mgc.axis_points(count))
        for i, f in enumerate(mgc.asym_models(ns, count)):
            ref = g['v_%d_%d' % (count, i)]
            o = oracle_lib.evaluate(f, P)
            assert np.all(np.abs(o - ref) <= value_tolerance(ref, P)), (count, i, float(np.max(np.abs(o - ref))))

