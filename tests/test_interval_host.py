"""The interval primitives behind the per-batch tape pruning and the cell-group culling
(sdf_amd/csrc/sdf_interval.h) are __host__ __device__: this builds them into a CPU program
(tests/native/interval_host.hip, host side only) that checks, on random boxes, that every value
computed at a point of the box lies inside the interval computed for the box -- circular_array
(hypot / atan2 / floored modulo / sin / cos), repeat (rint / clip), the monotone easing curves and
the interval product."""
import os
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_interval_primitives_enclose_sampled_points(tmp_path):
    exe = str(tmp_path / 'interval_host')
    subprocess.check_call([HIPCC, '--offload-host-only', '-O1', '-std=c++17', '-ffp-contract=off', '-w',
                           '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', exe,
                           os.path.join(ROOT, 'tests', 'native', 'interval_host.hip')])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-4000:]
    assert 'fails 0' in out.stdout


def _tape_lib(tmp_path_factory):
    import ctypes
    so = str(tmp_path_factory.mktemp('ia') / 'libia_tape.so')
    subprocess.check_call([HIPCC, '--offload-host-only', '-O1', '-std=c++17', '-ffp-contract=off', '-w', '-fPIC', '-shared',
                           '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', so,
                           os.path.join(ROOT, 'tests', 'native', 'interval_tape_host.hip')])
    lib = ctypes.CDLL(so)
    lib.ia_tape_boxes.restype = ctypes.c_int
    lib.ia_tape_boxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    return lib


@pytest.fixture(scope='module')
def tape_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not installed')
    return _tape_lib(tmp_path_factory)


def _fixture_names():
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fixtures
    return sorted(fixtures.FIXTURES)


def _check_tape_enclosure(name, f, tape_lib, oracle_lib, centres, seed, nb=1500, tight=False):
    """boxes of many sizes around `centres`: the checker's values at the corners and at random points of each box must
    lie inside the interval the host build of ia_run_tape computes for the box"""
    import numpy as np
    from sdf_amd import tape as tape_mod
    t = tape_mod.lower(f)
    rng = np.random.default_rng(seed)
    c = centres[rng.integers(0, len(centres), nb)].astype(np.float64)
    if t.dim == 2:
        c[:, 2] = 0.0
    size = 10.0 ** rng.uniform(-3.5, 0.0, (nb, 1)) * rng.uniform(0.2, 1.0, (nb, 3))
    size[rng.random(nb) < 0.1] *= np.array([1.0, 0.0, 1.0])                       # some degenerate boxes
    lo, hi = c - size * rng.random((nb, 3)), c + size * rng.random((nb, 3))
    if t.dim == 2:
        lo[:, 2] = hi[:, 2] = 0.0
    boxes = np.ascontiguousarray(np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1], lo[:, 2], hi[:, 2]], axis=1))
    out = np.empty((nb, 2))
    code = np.ascontiguousarray(t.code, dtype=np.uint32)
    consts = np.ascontiguousarray(np.concatenate([t.consts, [0.0]]))
    assert tape_lib.ia_tape_boxes(code.ctypes.data, consts.ctypes.data, t.n_instr, t.n_pslots, t.n_dslots,
                                  boxes.ctypes.data, nb, out.ctypes.data) == 0
    assert not np.isnan(out).any() and np.all(out[:, 0] <= out[:, 1])
    k = 16                                                                         # 8 corners + 8 random points per box
    u = rng.random((nb, k, 3))
    u[:, :8, :] = np.array([[(q >> 2) & 1, (q >> 1) & 1, q & 1] for q in range(8)], dtype=np.float64)[None]
    pts = (lo[:, None, :] + (hi - lo)[:, None, :] * u).reshape(-1, 3)
    pts = np.minimum(np.maximum(pts, np.repeat(lo, k, axis=0)), np.repeat(hi, k, axis=0))
    v = oracle_lib.evaluate(f, pts if t.dim == 3 else np.ascontiguousarray(pts[:, :2])).reshape(nb, k)
    L, H = out[:, :1], out[:, 1:]
    bad = ~(((v >= L) & (v <= H)) | (np.isnan(v) & np.isinf(L) & np.isinf(H)))
    if bad.any():
        i, j = np.argwhere(bad)[0]
        raise AssertionError('%s: box %r point %r value %.17g not in [%.17g, %.17g]' % (name, boxes[i].tolist(), pts[i * k + j].tolist(), v[i, j], out[i, 0], out[i, 1]))
    if tight:      # not vacuous: the small boxes get finite intervals
        small = size.max(axis=1) < 1e-2
        assert np.isfinite(out[small]).all(axis=1).mean() > 0.9, name


@pytest.mark.parametrize('kind,seed', [(k, s) for k in ('csg', 'array', 'leaf') for s in range(12)])
def test_interval_run_of_random_trees_encloses_the_checker(kind, seed, tape_lib, ns, golden_values, oracle_lib):
    """the same on random trees: booleans (hard and smooth) of transformed leaves, arrays / bends / twists / transitions
    with random easings, and booleans of the composed leaves -- the generators of the GPU on / off identity tests"""
    import numpy as np
    import test_gpu as tg
    rng = np.random.default_rng(7000 + 100 * ('csg', 'array', 'leaf').index(kind) + seed)
    f = {'csg': tg._random_csg, 'array': tg._random_array_tree, 'leaf': tg._random_leaf_tree}[kind](rng, ns)
    _check_tape_enclosure('%s-%d' % (kind, seed), f, tape_lib, oracle_lib, golden_values['P'], seed + 1, nb=1000)


@pytest.mark.parametrize('name', _fixture_names())
def test_interval_run_of_every_fixture_tape_encloses_the_checker(name, tape_lib, ns, golden_values, oracle_lib):
    """The WHOLE interval interpreter (ia_run_tape: every op's interval form, the prefixes, the slots), built for the
    host, on the tape of every value fixture: boxes of many sizes around the golden sample points; the CPU checker's
    values at the corners and at random points of each box must lie inside the box's interval (a NaN value -- the
    reference produces some -- is only allowed where the interval is the whole line)."""
    import fixtures
    tight = name.startswith('ex_') or name in ('sphere', 'box2', 'torus', 'capsule', 'smooth_union', 'twist', 'bend', 'repeat3', 'circular_array')
    _check_tape_enclosure(name, fixtures.build(name, ns), tape_lib, oracle_lib, golden_values['P'], zlib.crc32(name.encode()), tight=tight)
