"""The tile's task map and k_cull's body on the host (tests/native/tile_tasks_host.py: the device source of `TileTasks` and
`cull_tasks` with the device idioms replaced, a workgroup played by host threads) against a NumPy restatement of their
rules.  No GPU.
* `TileTasks`: every sample of a tile -- the full 33^3 one (512 cubes of 4^3 samples + its three far faces) and ragged ones
  (runs of 64 consecutive samples) -- belongs to exactly one (task, lane);
* `cull_tasks`: two interval levels (8^3 boxes, then the 4^3 groups inside the undecided ones); a task is listed iff a cell
  its samples touch lies in an undecided group; every sample that belongs to an undecided group is in a listed task."""
import ctypes
import os
import sys

import numpy as np
import pytest

import fixtures
from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
import tile_tasks_host
import test_interval_host as tih


@pytest.fixture(scope='module')
def libs(tmp_path_factory):
    if not os.path.exists(tile_tasks_host.HIPCC):
        pytest.skip('hipcc not installed')
    return tile_tasks_host.build(str(tmp_path_factory.mktemp('tile'))), tih._tape_lib(tmp_path_factory)


def _task_samples(lib, n):
    """task -> list of (ix, iy, iz) by the device's map; asserts that idle lanes still return valid indices"""
    out = (ctypes.c_int * 3)()
    tasks = []
    for task in range(lib.tile_ntask(*n)):
        own = []
        for lane in range(64):
            ok = lib.tile_sample(n[0], n[1], n[2], task, lane, out)
            assert 0 <= out[0] < n[0] and 0 <= out[1] < n[1] and 0 <= out[2] < n[2]
            if ok:
                own.append((out[0], out[1], out[2]))
        tasks.append(own)
    return tasks


@pytest.mark.parametrize('shape', [(33, 33, 33), (32, 33, 33), (33, 33, 32), (33, 17, 5), (2, 2, 2), (7, 33, 33), (33, 2, 33), (19, 19, 19), (1, 33, 33)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_every_sample_of_a_tile_belongs_to_exactly_one_task_lane(shape, libs):
    lib, _ = libs
    seen = np.zeros(shape, np.int32)
    for own in _task_samples(lib, shape):
        for q in own:
            seen[q] += 1
    assert (seen == 1).all()
    if shape == (33, 33, 33):
        assert lib.tile_ntask(*shape) == 563
    else:
        assert lib.tile_ntask(*shape) == (int(np.prod(shape)) + 63) // 64


def _states(tape_lib, t, boxes):
    boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
    out = np.empty((len(boxes), 2))
    code = np.ascontiguousarray(t.code, dtype=np.uint32)
    consts = np.ascontiguousarray(np.concatenate([t.consts, [0.0]]))
    if len(boxes):
        assert tape_lib.ia_tape_boxes(code.ctypes.data, consts.ctypes.data, t.n_instr, t.n_pslots, t.n_dslots, boxes.ctypes.data, len(boxes), out.ctypes.data) == 0
    return np.where(out[:, 0] > 1e-30, 1, np.where(out[:, 1] < -1e-30, 2, 0)).astype(np.uint8)


def _group_states(tape_lib, t, ax):
    """8^3 group states by the rules: a group outside the tile counts as decided (1); a group inherits its 8^3-cell box's
    state; the groups of undecided boxes are evaluated themselves"""
    n = [len(a) for a in ax]
    c = [m - 1 for m in n]

    def level(size, count, parent):
        st = np.ones((count,) * 3, np.uint8)
        idx = [q for q in np.ndindex(count, count, count) if all(size * q[d] < c[d] for d in range(3))]
        if parent is not None:
            for q in idx:
                st[q] = parent[q[0] >> 1, q[1] >> 1, q[2] >> 1]
            idx = [q for q in idx if st[q] == 0]
        boxes = []
        for q in idx:
            b = []
            for d in range(3):
                lo, hi = ax[d][size * q[d]], ax[d][min(size * q[d] + size, c[d])]
                b += [min(lo, hi), max(lo, hi)]
            boxes.append(b)
        for q, r in zip(idx, _states(tape_lib, t, boxes)):
            st[q] = r
        return st
    return level(4, 8, level(8, 4, None))


TILES = [('ex_example', 2 ** 22, 'regular'), ('ex_example', 1500000, 'ragged'), ('ex_blobby', 2 ** 21, 'regular'), ('ex_blobby', 1500000, 'ragged'),
         ('ex_gearlike', 2 ** 21, 'regular'), ('ex_gearlike', 1200000, 'ragged'), ('ex_weave', 2 ** 22, 'regular'), ('ex_knurling', 2 ** 21, 'regular')]


@pytest.mark.parametrize('name,samples,kind', TILES, ids=['%s-%s' % (n[3:], k) for n, _, k in TILES])
def test_cull_tasks_on_the_host_match_the_rules(name, samples, kind, libs, ns):
    from sdf_amd import core, tape as tape_mod
    lib, tape_lib = libs
    f = fixtures.build(name, ns)
    bounds = np.load(os.path.join(GOLDEN, 'bounds.npz'))[name]
    X, Y, Z, _ = core.grid_axes(tuple(map(tuple, bounds)), samples=samples)
    t = tape_mod.lower(f)
    # the first batch of the wanted shape (a full 33^3 tile / a tile with a shorter axis) that the surface crosses
    ax = want_g = None
    nb = [-(-len(a) // 32) for a in (X, Y, Z)]
    for b in np.ndindex(*nb):
        cand = [a[32 * o: 32 * o + 33] for a, o in ((X, b[0]), (Y, b[1]), (Z, b[2]))]
        n = [len(a) for a in cand]
        if min(n) < 2 or (kind == 'regular') != (n == [33, 33, 33]):
            continue
        box = [v for a in cand for v in (min(a[0], a[-1]), max(a[0], a[-1]))]
        if _states(tape_lib, t, [box])[0] == 0:
            g = _group_states(tape_lib, t, cand)
            if 4 <= (g == 0).sum() < 200:
                ax, want_g = cand, g
                break
    assert ax is not None, 'no such batch on this grid'
    n = [len(a) for a in ax]
    c = [m - 1 for m in n]
    axes = np.zeros(99)
    for d in range(3):
        axes[33 * d:33 * d + n[d]] = ax[d]
    code = np.ascontiguousarray(t.code, dtype=np.uint32)
    consts = np.ascontiguousarray(np.concatenate([t.consts, [0.0]]))
    got = {}
    for block in (64, 128, 256):
        rec = np.zeros(lib.cull_record_bytes(), np.uint8)
        ntl = ctypes.c_int(0)
        assert lib.cull_host(block, code.ctypes.data, consts.ctypes.data, t.n_instr, t.n_pslots, t.n_dslots, n[0], n[1], n[2],
                             axes.ctypes.data, rec.ctypes.data, ctypes.byref(ntl)) == 0
        assert ntl.value >= 0
        tlist = rec[2:2 + 2 * ntl.value].view(np.uint16).copy()                   # (u16 0 of the record is the caller's: the count)
        gstate = rec[lib.cull_gstate_offset():lib.cull_gstate_offset() + 512].reshape(8, 8, 8).copy()
        got[block] = (ntl.value, tlist, gstate)
    ntl, tlist, gstate = got[256]
    for block in (64, 128):
        assert got[block][0] == ntl and np.array_equal(got[block][1], tlist) and np.array_equal(got[block][2], gstate)
    assert np.array_equal(gstate, want_g)
    # the rule: a task is listed iff one of the cells its samples touch (the cells i - 1 and i around sample i, per axis; for
    # a task that is a run of samples: the rows / planes the run spans) lies in an undecided group -- restated on the task map
    und = want_g == 0

    def unknown_in(lo, hi):
        sl = []
        for d in range(3):
            a, b = max(lo[d], 0), min(hi[d], c[d] - 1)
            if b < a:
                return False
            sl.append(slice(a >> 2, (b >> 2) + 1))
        return bool(und[tuple(sl)].any())
    tasks = _task_samples(lib, n)
    want_list = []
    for task, own in enumerate(tasks):
        q = np.array(own)
        lo, hi = q.min(axis=0), q.max(axis=0)
        if n == [33, 33, 33] and task < 512:
            need = unknown_in(lo - 1, hi)                                         # a cube of 4^3 samples
        elif n == [33, 33, 33]:
            # a run of 64 samples of a far face: the face's own coordinate is 32 (cell 31); along the run's slow in-face axis the
            # rows it spans, along the fast one everything
            if task < 530:
                need = unknown_in((31, lo[1] - 1, 0), (31, hi[1], 31))
            elif task < 547:
                need = unknown_in((lo[0] - 1, 31, 0), (hi[0], 31, 31))
            else:
                need = unknown_in((lo[0] - 1, 0, 31), (hi[0], 31, 31))
        else:
            # a run of 64 consecutive samples: one row (then the z range it covers), several rows of a plane (then whole rows),
            # or several planes (then whole planes)
            if lo[0] != hi[0]:
                need = unknown_in((lo[0] - 1, -1, -1), (hi[0], n[1] - 1, n[2] - 1))
            elif lo[1] != hi[1]:
                need = unknown_in((lo[0] - 1, lo[1] - 1, -1), (hi[0], hi[1], n[2] - 1))
            else:
                need = unknown_in(lo - 1, hi)
        if need:
            want_list.append(task)
    assert np.array_equal(tlist, np.array(want_list, np.uint16))
    assert 0 < ntl < len(tasks)
    # soundness: every sample that belongs to an undecided group is in a listed task
    listed = np.zeros(n, bool)
    for task in tlist:
        for q in tasks[int(task)]:
            listed[q] = True
    for gq in np.argwhere(und):
        sl = tuple(slice(4 * int(gq[d]), min(4 * int(gq[d]) + 4, c[d]) + 1) for d in range(3))
        assert listed[sl].all()
