"""Host-side logic that needs no GPU: the drop-in API surface, tape lowering, header sync,
the C ABI symbol table and the STL writer."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import fixtures
from conftest import ROOT, GOLDEN


def test_public_names_cover_reference_surface(ns):
    # the names `from sdf import *` gives in the reference (reference sdf/__init__.py:1-27;
    # SURVEY.md section 8b), minus imported helper modules
    expected = '''
    ORIGIN X Y Z UP pi degrees radians d2 d3 ease
    SDF2 SDF3 sdf2 sdf3 op2 op3 op23 op32
    sphere plane slab box rounded_box wireframe_box torus capsule cylinder capped_cylinder
    rounded_cylinder capped_cone rounded_cone ellipsoid pyramid tetrahedron octahedron
    dodecahedron icosahedron
    translate scale rotate rotate_to orient circular_array elongate twist bend bend_linear
    bend_radial transition_linear transition_radial wrap_around slice
    union difference intersection blend negate dilate erode shell repeat
    circle line rectangle rounded_rectangle equilateral_triangle hexagon rounded_x polygon vesica
    extrude extrude_to revolve
    Mesh measure_image measure_text image text
    generate save sample_slice show_slice write_binary_stl
    '''.split()
    missing = [n for n in expected if n not in ns]
    assert not missing, missing


def test_alias_package_is_drop_in():
    import sdf
    import sdf_amd
    assert sdf.sphere is sdf_amd.sphere and sdf.d3 is sdf_amd.d3 and sdf.ease is sdf_amd.ease
    assert sdf.X.tolist() == [1, 0, 0] and sdf.UP is sdf.Z      # d3 names shadow d2's


def test_k_quirks(ns):
    """reference quirks listed in SURVEY.md A.3"""
    cyl, Xv, Zv = ns['cylinder'], ns['X'], ns['Z']
    c = cyl(0.5)
    assert c.k(0.1) is c and c._k == 0.1                         # .k() mutates and returns self
    assert cyl(.5).k(.1).orient(Zv)._k == 0.1                    # parallel -> child itself, _k visible
    assert getattr(cyl(.5).k(.1).orient(Xv), '_k', None) is None
    assert getattr(ns['circle'](1), '_k', None) is None          # SDF2 has no fall-through


def test_smoothing_constant_is_resolved_at_evaluation_time(ns):
    from sdf_amd import tape
    a, b = ns['sphere'](1), ns['box'](1.5)
    f = a | b
    assert 'SUNION' not in tape.lower(f).disassemble()
    b.k(0.25)                                                    # after building the union
    assert 'SUNION' in tape.lower(f).disassemble()


@pytest.mark.parametrize('name', sorted(fixtures.FIXTURES))
def test_every_fixture_lowers(name, ns):
    from sdf_amd import tape
    t = tape.lower(fixtures.build(name, ns))
    assert t.n_instr >= 2 and t.code.dtype == np.uint32 and t.consts.dtype == np.float64
    assert (t.code[-2] & 255) == tape.OP['END']
    assert t.n_pslots <= tape.MAX_P_SLOTS and t.n_dslots <= tape.MAX_D_SLOTS


def test_user_closures_lower_to_extern_leaves(ns):
    """a function decorated with @sdf3 / @op3 that returns NumPy code (reference README.md:258-295) becomes an
    `extern` leaf: the tape remembers the callable, the device reads its values from a buffer"""
    from sdf_amd import tape, ir

    @ns['sdf3']
    def custom():
        def f(p):
            return np.linalg.norm(p, axis=1) - 1
        return f
    c = custom()
    t = tape.lower(c | ns['sphere'](1))
    assert 'L_EXTERN' in t.disassemble() and len(t.externs) == 1 and t.externs[0][1] == 3
    assert t.rstart is None                                   # no interval prepass around host code
    t2 = tape.lower(c.translate((1, 0, 0)) | c)               # one closure, two leaves
    assert len(t2.externs) == 2 and t2.externs[0][0] is t2.externs[1][0]
    with pytest.raises(ir.OpaqueSDFError):
        tape.lower(ns['SDF3'](42) | ns['sphere'](1))          # neither a node nor a callable
    for name in fixtures.CUSTOM_FIXTURES:
        t = tape.lower(fixtures.build(name, ns))
        assert len(t.externs) >= 1 and (t.code[-2] & 255) == tape.OP['END']


def test_grid_leaf_matches_the_reference_closure_on_the_oracle(ns, oracle_lib):
    """reference sdf/mesh.py:96-105 (scipy RegularGridInterpolator + box estimator) restated in the oracle:
    bit-identical to values produced by the reference's own ingredients (tools/make_golden_custom.py)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import make_golden_custom as mgc
    from sdf_amd import mesh, tape
    g = np.load(os.path.join(GOLDEN, 'grid3d.npz'))
    for name, (X, Y, Z, A, bg, bb) in mgc.grids().items():
        f = mesh.grid_sdf((X, Y, Z), A, bg, bb)
        assert np.array_equal(oracle_lib.evaluate(f, g['p_' + name]), g['v_' + name]), name
        h = f.translate((0.05, -0.03, 0.02)) | ns['sphere'](0.2).translate((0, 0, 0.5))
        assert np.array_equal(oracle_lib.evaluate(h, g['p_' + name]), g['vc_' + name]), name
        assert 'L_GRID3D' in tape.lower(h).disassemble()
    with pytest.raises(ValueError):
        mesh.grid_sdf((np.arange(3.0), np.arange(3.0), np.array([0.0, 2.0, 1.0])), np.zeros((3, 3, 3)), 1.0, ((0, 0, 0), (1, 1, 1)))
    m = mesh.Mesh(np.array([[0.0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]]), np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]]))
    assert m.size == (1.0, 2.0, 3.0) and m.centered().bounding_box == ((-0.5, -1.0, -1.5), (0.5, 1.0, 1.5))
    assert m.scaled(2).translated((1, 1, 1)).bounding_box == ((1.0, 1.0, 1.0), (3.0, 5.0, 7.0))
    with pytest.raises(ImportError):                          # the voxeliser is OpenVDB's, exactly like the reference
        m.sdf(0.1)


def test_generated_headers_are_in_sync():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_headers
    for rel, text in gen_headers.render().items():
        assert open(os.path.join(ROOT, rel)).read() == text, rel + ' is stale: run tools/gen_headers.py'
    a = open(os.path.join(ROOT, 'oracle', 'mc_table.h')).read()
    b = open(os.path.join(ROOT, 'sdf_amd', 'csrc', 'mc_table.h')).read()
    assert a == b


def test_c_abi_exports_every_declared_symbol():
    """libsdf_hip.so loads without a GPU and exports exactly what include/sdf_hip.h declares"""
    from sdf_amd import engine
    hdr = open(os.path.join(ROOT, 'include', 'sdf_hip.h')).read()
    declared = set(re.findall(r'\b(sdf_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(engine.ABI), declared ^ set(engine.ABI)
    lib = engine.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sdf_abi_version() == engine.ABI_VERSION
    # struct layout of sdf_stats: 11 int64 + 4 double + 3 int64 + 4 double + 1 int64 (ABI 8: mesh_kernel); sdf_exchange_stats: 14 int64 + 4 double + 64 int64
    assert ctypes.sizeof(engine.SdfStats) == 23 * 8
    assert ctypes.sizeof(engine.SdfExchangeStats) == (14 + 4 + 64) * 8


def test_no_cpu_fallback_without_device(ns):
    """on a box without a GPU every compute entry point raises instead of silently
    computing on the host"""
    from sdf_amd import engine
    lib = engine.load_library()
    if lib.sdf_device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(engine.SdfHipError):
        ns['sphere'](1)(np.zeros((1, 3)))
    with pytest.raises(engine.SdfHipError):
        ns['sphere'](1).generate(samples=2 ** 12, verbose=False)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'sdf_amd')):
        for fn in files:
            if fn.endswith(('.py', '.h', '.hip', '.cpp')):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', src, re.M), fn
                assert 'libsdf_oracle' not in src, fn


def test_stl_writer_bytes_match_reference(tmp_path):
    from sdf_amd import stl
    pts = np.load(os.path.join(GOLDEN, 'gen_example_s15.npz'))['points']
    ref = np.load(os.path.join(GOLDEN, 'stl_example_s15.npz'))['stl'].tobytes()
    p = str(tmp_path / 'a.stl')
    stl.write_binary_stl(p, pts)
    assert open(p, 'rb').read() == ref
    stl.write_binary_stl(p, list(pts))           # the reference passes a list of points
    assert open(p, 'rb').read() == ref


def test_grid_axes_follow_reference_rule():
    from sdf_amd import core
    b = ((-0.845430, -0.845430, -0.845430), (0.845431, 0.845431, 0.845431))
    X, Y, Z, (dx, dy, dz) = core.grid_axes(b, samples=2 ** 22)
    assert len(X) == len(Y) == len(Z) == 162 and dx == dy == dz      # BASELINE config 1: 162^3
    X, Y, Z, _ = core.grid_axes(b, step=(0.1, 0.2, 0.4))
    assert (len(X), len(Y), len(Z)) == (17, 9, 5)


def test_shard_bounds_partition_the_work_list():
    from sdf_amd import dist
    for n in (0, 1, 7, 1744, 33856):
        for world in (1, 2, 3, 8):
            cuts = [dist.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def test_pinned_result_buffers_fall_back_to_pageable_memory_without_a_device():
    """results land in the library's pinned blocks on a GPU box (tests/test_gpu.py); where pinning is not
    possible the same call hands out an ordinary array"""
    import numpy as np
    from sdf_amd import engine
    lib = engine.load_library()
    small = engine.pinned_empty(lib, (10, 3), np.float64)              # below 1 MiB: never pinned
    assert small.shape == (10, 3) and small.base is None
    big = engine.pinned_empty(lib, (1 << 18, 3), np.float64)           # 6 MiB
    assert big.shape == (1 << 18, 3) and big.dtype == np.float64
    big[:] = 1.0                                                       # writable either way
    assert float(big.sum()) == 3.0 * (1 << 18)


def test_long_tapes_carry_no_prune_info_and_big_pools_fail_cleanly(ns):
    """operand ranges are 16-bit and only used for tapes of <= 256 instructions -- longer tapes must lower
    without them (they ran into an OverflowError beyond 65535 instructions); a constant pool beyond the 24-bit offset
    of an instruction is a ValueError, not a bare assert"""
    from sdf_amd import tape
    parts = [ns['sphere'](0.1).translate((0.01 * i, 0, 0)) for i in range(200)]
    t = tape.lower(ns['union'](*parts))
    assert t.n_instr > tape.PRUNE_MAX_INSTR and t.rstart is None and t.lstart is None
    small = tape.lower(ns['union'](*parts[:20]))
    assert small.rstart is not None and len(small.rstart) == small.n_instr
    lw = tape._Lowering()
    lw.consts.extend([0.0] * (tape.COFF_MASK - 6))
    lw.emit('L_SPHERE', consts=(1.0, 0.0, 0.0, 0.0))            # still addressable
    with pytest.raises(ValueError):
        lw.emit('L_SPHERE', consts=(1.0, 0.0, 0.0, 0.0))


def test_bound_method_closures_keep_their_own_instance():
    """two bound methods of one function on different instances are two closures (a bound method's __dict__ is the
    function's: a node cached there evaluated the first instance's closure for both)"""
    from sdf_amd import ir

    class Ball:
        def __init__(self, r):
            self.r = r

        def f(self, p):
            return np.linalg.norm(p, axis=1) - self.r

    a, b = Ball(1.0), Ball(2.0)
    na, nb = ir.extern_node(a.f), ir.extern_node(b.f)
    assert na is not nb and na.meta['fn'].__self__ is a and nb.meta['fn'].__self__ is b
    assert ir.extern_node(a.f) is na and ir.extern_node(b.f) is nb      # one node per (instance, function)
    P = np.array([[3.0, 0.0, 0.0]])
    assert na.meta['fn'](P)[0] == 2.0 and nb.meta['fn'](P)[0] == 1.0

    class Slotted:
        __slots__ = ('r',)

        def f(self, p):
            return np.linalg.norm(p, axis=1) - self.r

    s1, s2 = Slotted(), Slotted()
    s1.r, s2.r = 1.0, 2.0
    assert ir.extern_node(s1.f).meta['fn'].__self__ is s1 and ir.extern_node(s2.f).meta['fn'].__self__ is s2

    def plain(p):
        return p[:, 0]
    assert ir.extern_node(plain) is ir.extern_node(plain)


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` started plainly launches its own ranks -- after checking that N devices are visible;
    on a box with fewer (none, here) it says so instead of starting ranks that cannot get a GPU"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SDF_BENCH_ONE_DEVICE')}
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
            pytest.skip('64 devices visible')
    except ImportError:
        pytest.skip('no torch')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'HIP device(s) visible' in (r.stderr + r.stdout)
    # ... and a rank that finds itself in a world of another size than it was told refuses as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=dict(env, RANK='0', WORLD_SIZE='3', LOCAL_RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=3' in (r.stderr + r.stdout)


def test_build_self_check_and_build_info():
    """tools/isa_check.py (run by __graft_entry__.build()): the tape interpreters of the built library carry their scalar
    jump-table dispatch, the kernels of the plain translation units do not, every kernel the host launches is there; and the
    library says which toolchain built it (sdf_build_info)"""
    import subprocess
    import sys
    from sdf_amd import engine
    engine.load_library()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_check.py'), engine.LIB_PATH], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert 'k_mesh<double, false, 1, 1, 3, 1024, false>' in r.stdout and 'ISA CHECK FAILED' not in r.stdout
    info = engine.build_info()
    assert 'HIP version' in info and 'structurizecfg-skip-uniform-regions' in info, info


def test_source_id_ignores_comments_not_string_literals():
    """engine.source_id() hashes the sources without their comments; a `//` inside a string or character literal is not one"""
    from sdf_amd import engine
    strip = engine._strip_c_comments
    assert strip('a = "http://x"; // c1\n/* c2 */ b = \'"\'; c = "\\"//"; // end\nd') == 'a = "http://x"; \n b = \'"\'; c = "\\"//"; \nd'
    assert strip('x /* a // b */ y // z') == 'x  y '
    assert len(engine.source_id()) == 16


def test_bench_whole_soup_comparison_counts_differences(ns):
    """bench.py's `whole_soup_vs_oracle` (every coordinate of a config's soup against the checker meshing the grid in processes of
    their own): a soup that IS the checker's differs nowhere; one perturbed coordinate is counted, with its size"""
    import subprocess
    import sys
    script = '''
import sys
sys.path.insert(0, %r)
import numpy as np, bench, oracle
from sdf_amd import core
if __name__ == '__main__':
    f, _ = bench.build_model('gearlike')
    bounds = ((-2.1, -2.1, -0.6), (2.1, 2.1, 0.6))
    X, Y, Z, _ = core.grid_axes(bounds, samples=2 ** 19)
    nb = (-(-len(X) // 32)) * (-(-len(Y) // 32)) * (-(-len(Z) // 32))
    ref = oracle.generate(f, X, Y, Z, 32, True).points
    offs = np.zeros(nb + 1, np.int64)          # where each batch's triangles start (the device's Mesh.batch_offsets())
    for b in range(nb):
        offs[b + 1] = offs[b] + len(oracle.generate(f, X, Y, Z, 32, True, batch_range=(b, b + 1)).points) // 3
    assert 3 * offs[-1] == len(ref)
    a = bench.whole_soup_vs_oracle('gearlike', bounds, 19, ref, offs, budget_cores=3)
    host = ref.copy(); host[len(host) // 2, 1] += 1e-7
    b = bench.whole_soup_vs_oracle('gearlike', bounds, 19, host, offs, budget_cores=3)
    assert a['coordinates_that_differ'] == 0 and a['whole_soup'] and a['within_1e-5'] and a['coordinates'] == 3 * len(ref) and a['coverage'] == 1.0, a
    assert b['coordinates_that_differ'] == 1 and abs(b['max_abs_diff_over_extent'] * 4.2 - 1e-7) < 1e-12 and b['whole_soup'], b
    # out of time before a piece came back: no verdict, coverage 0 (a comparison of nothing must not say "true")
    c = bench.soup_verdict(0, 3 * len(ref), 0, 0.0)
    assert c['coverage'] == 0.0 and c['within_1e-5'] is None and c['share_bit_equal'] is None and not c['whole_soup'], c
    # a soup with a triangle too many in one batch is an error, not a shifted comparison
    offs2 = offs.copy(); offs2[nb // 2 + 1:] += 1
    d = bench.whole_soup_vs_oracle('gearlike', bounds, 19, np.concatenate([ref, ref[:3]]), offs2, budget_cores=3)
    assert 'error' in d, d
    print('ok', a['coordinates'])
''' % ROOT
    path = os.path.join(ROOT, 'tests', '_whole_soup_check.py')
    try:
        open(path, 'w').write(script)         # (a real file: the spawned workers import the main module)
        r = subprocess.run([sys.executable, path], capture_output=True, text=True, timeout=600)
    finally:
        os.remove(path)
    assert r.returncode == 0 and r.stdout.strip().startswith('ok'), r.stdout[-1500:] + r.stderr[-3000:]



def test_bench_optional_section_cannot_swallow_the_headline_line():
    """bench.py runs its optional `other_configs` section under a watchdog -- on ONE GPU too (r05ae: a default run sat in that section
    until its caller's limit, no line): when the section does not come back, the line that is already complete is printed, with the
    reason in it, and the process exits with code 0"""
    import json
    import subprocess
    import sys
    script = ("import os, sys, time; sys.path.insert(0, %r); import bench\n"
              "out = {'metric': 'm', 'value': 1.5, 'other_configs': None}\n"
              "bench.watchdog(out)\n"
              "time.sleep(20)\n"
              "print('NOT REACHED')\n") % ROOT
    env = dict(os.environ, SDF_BENCH_OTHER_TIMEOUT_S='0.3')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and 'NOT REACHED' not in r.stdout, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] == 1.5 and line['stalled'] == ['other_configs'] and 'did not finish' in line['other_configs'][0]['error']


def test_bench_whole_soup_comparison_survives_dead_workers():
    """whole_soup_vs_oracle's workers are plain processes that leave files: workers that die (here: a model they cannot build) make the
    comparison stop with what it has (coverage 0, no verdict) -- at once, not at the end of its time budget, and never in a pool's shutdown"""
    import subprocess
    import sys
    import time
    script = '''
import sys
sys.path.insert(0, %r)
import numpy as np, bench
if __name__ == '__main__':
    r = bench.whole_soup_vs_oracle('no_such_model', ((-1, -1, -1), (1, 1, 1)), 15, np.zeros((30, 3)), np.arange(9) * 10 // 8, budget_cores=3, budget_s=120.0)
    assert r['coordinates'] == 0 and r['coverage'] == 0.0 and not r['whole_soup'] and r['within_1e-5'] is None and r['coordinates_that_differ'] is None, r
    print('ok', r['checker_seconds'])
''' % ROOT
    path = os.path.join(ROOT, 'tests', '_whole_soup_dead.py')
    t0 = time.time()
    try:
        open(path, 'w').write(script)
        r = subprocess.run([sys.executable, path], capture_output=True, text=True, timeout=100)
    finally:
        os.remove(path)
    assert r.returncode == 0 and r.stdout.strip().startswith('ok'), r.stdout[-1500:] + r.stderr[-3000:]
    assert time.time() - t0 < 60
