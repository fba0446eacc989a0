"""k_cull's body on the host (tests/native/cull_tasks_host.py: the device source of `cull_tasks` / `cull_sample` with the
device idioms replaced, a workgroup played by host threads) against a NumPy restatement of its rules: three interval
levels (8^3 boxes, 4^3 groups, 2^3 sub-groups), units of 2^3 samples listed iff one of the sub-groups {u - 1, u}^3 is
undecided, eight units to a task.  No GPU."""
import ctypes
import os
import sys

import numpy as np
import pytest

import fixtures
from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import cull_tasks_host
import test_interval_host as tih


@pytest.fixture(scope='module')
def libs(tmp_path_factory):
    if not os.path.exists(cull_tasks_host.HIPCC):
        pytest.skip('hipcc not installed')
    return cull_tasks_host.build(str(tmp_path_factory.mktemp('cull'))), tih._tape_lib(tmp_path_factory)


def _intervals(tape_lib, t, boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
    out = np.empty((len(boxes), 2))
    code = np.ascontiguousarray(t.code, dtype=np.uint32)
    consts = np.ascontiguousarray(np.concatenate([t.consts, [0.0]]))
    if len(boxes):
        assert tape_lib.ia_tape_boxes(code.ctypes.data, consts.ctypes.data, t.n_instr, t.n_pslots, t.n_dslots, boxes.ctypes.data, len(boxes), out.ctypes.data) == 0
    return np.where(out[:, 0] > 1e-30, 1, np.where(out[:, 1] < -1e-30, 2, 0)).astype(np.uint8)


def _model(tape_lib, t, ax, levels=3):
    """sub-group states (16^3, non-existent = 1) and the listed units (packed, ascending) by the rules, level by level
    (two levels: the sub-groups keep the states of their groups)"""
    n = [len(a) for a in ax]
    c = [m - 1 for m in n]

    def level(size, count, parent, evaluate=True):
        st = np.ones((count,) * 3, np.uint8)
        idx = [(i, j, k) for i in range(count) for j in range(count) for k in range(count)
               if size * i < c[0] and size * j < c[1] and size * k < c[2]]
        if parent is not None:
            for q in idx:
                st[q] = parent[q[0] >> 1, q[1] >> 1, q[2] >> 1]
            idx = [q for q in idx if st[q] == 0]
        if not evaluate:
            return st
        boxes = []
        for q in idx:
            b = []
            for d in range(3):
                lo, hi = ax[d][size * q[d]], ax[d][min(size * q[d] + size, c[d])]
                b += [min(lo, hi), max(lo, hi)]
            boxes.append(b)
        res = _intervals(tape_lib, t, boxes)
        for q, r in zip(idx, res):
            st[q] = r
        return st
    m = level(8, 4, None)
    g = level(4, 8, m)
    s = level(2, 16, g, evaluate=levels >= 3)
    nu = [(m_ + 1) >> 1 for m_ in n]
    units = []
    for u0 in range(nu[0]):
        for u1 in range(nu[1]):
            for u2 in range(nu[2]):
                sl = tuple(slice(max(u - 1, 0), min(u, 15) + 1) for u in (u0, u1, u2))
                if (s[sl] == 0).any():
                    units.append((u0 << 10) | (u1 << 5) | u2)
    return s, np.array(units, np.uint16)


def _run(cull_lib, t, ax, block, levels=3):
    lay = (ctypes.c_int * 4)()
    cull_lib.cull_layout(lay)
    ulist_off, sstate_off, cap = lay[0], lay[1], lay[2]
    rec = np.zeros(cull_lib.cull_record_bytes(), np.uint8)
    axes = np.zeros(99)
    for d in range(3):
        axes[33 * d:33 * d + len(ax[d])] = ax[d]
    ntl = ctypes.c_int(0)
    code = np.ascontiguousarray(t.code, dtype=np.uint32)
    consts = np.ascontiguousarray(np.concatenate([t.consts, [0.0]]))
    rc = cull_lib.cull_host(block, code.ctypes.data, consts.ctypes.data, t.n_instr, t.n_pslots, t.n_dslots, len(ax[0]), len(ax[1]), len(ax[2]),
                            axes.ctypes.data, rec.ctypes.data, ctypes.byref(ntl), levels)
    assert rc == 0
    n = int(rec[:2].view(np.uint16)[0])
    units = rec[ulist_off:ulist_off + 2 * cap].view(np.uint16)
    packed = rec[sstate_off:sstate_off + 1024].view(np.uint32)            # two bits per sub-group, 16 along h2 per word
    sstate = ((packed[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).astype(np.uint8).reshape(16, 16, 16)
    return ntl.value, n, units, sstate, rec


def _colinfo(cull_lib, rec):
    lay = (ctypes.c_int * 4)()
    cull_lib.cull_layout(lay)
    return rec[lay[3]:lay[3] + 4 * 289].view(np.uint32)


TILES = [('ex_example', 2 ** 22, 'regular'), ('ex_example', 1500000, 'ragged'), ('ex_blobby', 2 ** 21, 'regular'), ('ex_blobby', 1500000, 'ragged'),
         ('ex_gearlike', 2 ** 21, 'regular'), ('ex_gearlike', 1200000, 'ragged'), ('ex_weave', 2 ** 22, 'regular'), ('ex_knurling', 2 ** 21, 'regular')]


@pytest.mark.parametrize('levels', [3, 2])
@pytest.mark.parametrize('name,samples,kind', TILES, ids=['%s-%s' % (n[3:], k) for n, _, k in TILES])
def test_cull_tasks_on_the_host_match_the_rules(name, samples, kind, levels, libs, ns):
    from sdf_amd import core, tape as tape_mod
    cull_lib, tape_lib = libs
    f = fixtures.build(name, ns)
    bounds = np.load(os.path.join(GOLDEN, 'bounds.npz'))[name]
    X, Y, Z, _ = core.grid_axes(tuple(map(tuple, bounds)), samples=samples)
    t = tape_mod.lower(f)
    # the first batch of the wanted shape (a full 33^3 tile / a tile with a shorter axis) that the surface crosses
    ax = None
    nb = [-(-len(a) // 32) for a in (X, Y, Z)]
    for b in np.ndindex(*nb):
        cand = [a[32 * o: 32 * o + 33] for a, o in ((X, b[0]), (Y, b[1]), (Z, b[2]))]
        n = [len(a) for a in cand]
        if min(n) < 2 or (kind == 'regular') != (n == [33, 33, 33]):
            continue
        box = [v for a in cand for v in (min(a[0], a[-1]), max(a[0], a[-1]))]
        if _intervals(tape_lib, t, [box])[0] == 0:
            _, wu = _model(tape_lib, t, cand)
            if 0 < len(wu) < np.prod([(m + 1) >> 1 for m in n]) // 2:
                ax = cand
                break
    assert ax is not None, 'no such batch on this grid'
    n = [len(a) for a in ax]
    t = tape_mod.lower(f)
    want_s, want_units = _model(tape_lib, t, ax, levels)
    got = {}
    for block in (64, 128, 256):
        ntl, cnt, units, sstate, rec = _run(cull_lib, t, ax, block, levels)
        got[block] = (ntl, cnt, units[:((cnt + 7) & ~7) if cnt != 0xFFFF else 0].copy(), sstate.copy(), rec.copy())
    ntl, cnt, units, sstate, rec = got[256]
    for block in (64, 128):
        assert got[block][0] == ntl and got[block][1] == cnt and np.array_equal(got[block][2], units) and np.array_equal(got[block][3], sstate)
    assert cnt != 0xFFFF and ntl == (cnt + 7) >> 3
    assert np.array_equal(sstate, want_s)
    assert np.array_equal(units[:cnt], want_units)
    assert (units[cnt:] == 0xFFFF).all() and len(units) == 8 * ntl
    # how a lane finds its sample: every sample of every listed unit exactly once, nothing outside the tile
    seen = np.zeros(n, np.int32)
    out = (ctypes.c_int * 3)()
    ubuf = np.ascontiguousarray(units)
    for task in range(ntl):
        for lane in range(64):
            if cull_lib.cull_sample_host(ubuf.ctypes.data, task, lane, n[0], n[1], n[2], out):
                seen[out[0], out[1], out[2]] += 1
            else:
                assert (out[0], out[1], out[2]) == (0, 0, 0)
    assert seen.max() <= 1
    listed = np.zeros(n, bool)
    for u in want_units:
        u0, u1, u2 = int(u) >> 10, (int(u) >> 5) & 31, int(u) & 31
        listed[2 * u0:2 * u0 + 2, 2 * u1:2 * u1 + 2, 2 * u2:2 * u2 + 2] = True
    assert np.array_equal(seen.astype(bool), listed)
    # k_mesh's sign fill: a sample's bit is set iff the sub-group that owns it (min(i, c - 1) >> 1 per axis) is decided positive
    nvox = n[0] * n[1] * n[2]
    bits = np.zeros((nvox + 63) // 64 + 2, np.uint64)
    lay = (ctypes.c_int * 4)()
    cull_lib.cull_layout(lay)
    ss = np.ascontiguousarray(rec[lay[1]:lay[1] + 1024])                  # (the record's packed states, as k_mesh reads them)
    assert cull_lib.cull_sign_fill_host(ss.ctypes.data, n[0], n[1], n[2], bits.ctypes.data) == 0
    got_bits = np.unpackbits(bits.view(np.uint8), bitorder='little')[:nvox].reshape(n).astype(bool)
    cc = [m - 1 for m in n]
    own = [np.minimum(np.arange(n[d]), cc[d] - 1) >> 1 for d in range(3)]
    assert np.array_equal(got_bits, want_s[np.ix_(own[0], own[1], own[2])] == 1)
    assert not np.unpackbits(bits.view(np.uint8), bitorder='little')[nvox:].any()
    # k_mesh's sampling loop on the record: exactly the listed samples get their value (here: a code of their coordinates),
    # and the positive ones their sign bit on top of the fill
    idx_axes = np.zeros(99)
    for d_ in range(3):
        idx_axes[33 * d_:33 * d_ + n[d_]] = np.arange(n[d_])
    vol = np.full(nvox, np.float32(-12345.0), np.float32)
    bits2 = bits.copy()
    assert cull_lib.cull_sample_loop_host(np.ascontiguousarray(rec).ctypes.data, n[0], n[1], n[2], idx_axes.ctypes.data, vol.ctypes.data, bits2.ctypes.data, 0, None) == 0
    I, J, K = np.meshgrid(np.arange(n[0]), np.arange(n[1]), np.arange(n[2]), indexing='ij')
    code_of = (I + 64.0 * J + 4096.0 * K - 70000.0).astype(np.float32)
    vol3 = vol.reshape(n)
    assert np.array_equal(vol3[listed], code_of[listed]) and (vol3[~listed] == np.float32(-12345.0)).all()
    want_bits2 = got_bits | (listed & (code_of > 0))
    assert np.array_equal(np.unpackbits(bits2.view(np.uint8), bitorder='little')[:nvox].reshape(n).astype(bool), want_bits2)
    # the same loop on a SPARSE tile (k_mesh keeps only the listed units' samples, 64 floats per task): sample `lane` of
    # task t sits at smp[64 t + lane], and TileView::at finds every listed sample there through the column words
    smp = np.full(64 * ntl, np.float32(-777.0), np.float32)
    vol_s = np.full(nvox, np.float32(-12345.0), np.float32)
    bits3 = bits.copy()
    assert cull_lib.cull_sample_loop_host(np.ascontiguousarray(rec).ctypes.data, n[0], n[1], n[2], idx_axes.ctypes.data, vol_s.ctypes.data, bits3.ctypes.data,
                                          1, smp.ctypes.data) == 0
    assert (vol_s == np.float32(-12345.0)).all() and np.array_equal(bits3, bits2)
    colinfo = np.ascontiguousarray(_colinfo(cull_lib, rec))
    nu = [(m_ + 1) >> 1 for m_ in n]
    k = 0
    for u0 in range(17):               # the column words: listed u2 | index of the column's first listed unit << 17
        for u1 in range(17):
            mine = [int(u) & 31 for u in want_units if (int(u) >> 10, (int(u) >> 5) & 31) == (u0, u1)]
            word = int(colinfo[u0 * 17 + u1])
            assert word & 0x1FFFF == sum(1 << u2 for u2 in mine), (u0, u1)
            if mine:
                assert word >> 17 == k
            k += len(mine)
    for (ix, iy, iz) in np.argwhere(listed)[::7]:
        got_v = cull_lib.tile_at_host(smp.ctypes.data, colinfo.ctypes.data, int(ix), int(iy), int(iz))
        assert got_v == code_of[ix, iy, iz], (ix, iy, iz)
    # soundness: every sample that belongs to an undecided sub-group is evaluated
    c = [m - 1 for m in n]
    for h in np.argwhere(want_s == 0):
        sl = tuple(slice(2 * int(h[d]), min(2 * int(h[d]) + 2, c[d]) + 1) for d in range(3))
        assert listed[sl].all()


def test_a_tile_with_nearly_everything_undecided_is_not_culled(libs, ns):
    """more units than the record holds: the dense path (return -1; the caller writes 0xFFFF into the header)"""
    from sdf_amd import tape as tape_mod
    cull_lib, tape_lib = libs
    f = ns['sphere'](0.003).repeat(0.01)                # a small sphere in every cell of the tile below
    t = tape_mod.lower(f)
    ax = [np.linspace(0.0, 0.32, 33)] * 3
    want_s, want_units = _model(tape_lib, t, ax)
    assert len(want_units) > 3072
    ntl, cnt, units, sstate, rec = _run(cull_lib, t, ax, 128)
    assert ntl == -1


def test_sign_fill_and_units_on_flat_tiles(libs, ns):
    """tiles with a short axis (lz = 5: two sub-groups along z, the second owning the boundary sample; non-existent
    sub-groups count as decided and must leave no bit) -- the grid of test_gpu.py::test_edge_cases"""
    from sdf_amd import core, tape as tape_mod
    cull_lib, tape_lib = libs
    f = fixtures.build('ex_example', ns)
    X, Y, Z, _ = core.grid_axes(((-0.9, -0.4, -0.2), (0.9, 0.5, 0.3)), (0.013, 0.05, 0.11))
    t = tape_mod.lower(f)
    lay = (ctypes.c_int * 4)()
    cull_lib.cull_layout(lay)
    seen = 0
    for bx in range(5):
        ax = [X[32 * bx:32 * bx + 33], Y[0:33], Z[0:33]]
        n = [len(a) for a in ax]
        for levels in (2, 3):
            want_s, want_units = _model(tape_lib, t, ax, levels)
            ntl, cnt, units, sstate, rec = _run(cull_lib, t, ax, 256, levels)
            assert cnt != 0xFFFF and np.array_equal(sstate, want_s) and np.array_equal(units[:cnt], want_units)
            nvox = n[0] * n[1] * n[2]
            bits = np.zeros((nvox + 63) // 64 + 2, np.uint64)
            ss = np.ascontiguousarray(rec[lay[1]:lay[1] + 1024])
            assert cull_lib.cull_sign_fill_host(ss.ctypes.data, n[0], n[1], n[2], bits.ctypes.data) == 0
            got = np.unpackbits(bits.view(np.uint8), bitorder='little')
            cc = [m - 1 for m in n]
            own = [np.minimum(np.arange(n[d]), cc[d] - 1) >> 1 for d in range(3)]
            assert np.array_equal(got[:nvox].reshape(n).astype(bool), want_s[np.ix_(own[0], own[1], own[2])] == 1)
            assert not got[nvox:].any()
            seen += int(cnt > 0)
    assert seen >= 4
