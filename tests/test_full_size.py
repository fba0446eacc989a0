"""Parity at the sizes BASELINE.json names: whole soups, not counts.

tests/golden/full_<tag>.npz were made by RUNNING the unmodified reference at these sizes
(tools/make_golden_full.py: bounds, step, per-batch classification, triangle count, sha256 of the
float64 soup and every 997th triangle of it).  The soups themselves are 0.2 - 0.7 GB each, so only
their hashes travel; the oracle (which finishes each configuration in 5 - 40 s on one core) produces
the full soup the HIP path is compared with row by row.

* CPU (`-m "not gpu"`): the oracle reproduces the reference's hash at full size (non-libm models).
* GPU: HIP soup == oracle soup == reference hash; the same with the interval passes switched off;
  for the libm models (device ocml vs glibc) the north-star tolerance 1e-5 x extent with > 99.9 % of
  the coordinates bit-equal, also against the reference's own sampled triangles.
* C4 (weave at 2**33, 266 256 batches; `generate` itself would take ~10 h and > 20 GB for its list of points): the
  reference's own per-batch function `_worker` was run over all of `generate`'s jobs in processes (tools/make_golden_c4.py ->
  full_c4_weave_s33.npz: classification and triangle count of EVERY batch, sha256, every 9973rd triangle) -- the HIP path
  must reproduce all of it; in addition >= 5 % of the surviving batches (seeded shuffle) are meshed by the oracle one by
  one and compared with the slices of the HIP soup that `sdf_mesh_batch_offsets` attributes to them.  On the CPU the
  oracle is held to the same file on the batches that own every 5th of the reference's sampled triangles.
* The multi-GPU shape of C2 / C3 / C4 / C5 on one device: emulated ranks mesh their shards into exchange slabs and
  `sdf_expand_slabs` must reproduce the reference's (resp. the single-GPU) soup hash.
"""
import hashlib
import os

import numpy as np
import pytest

import fixtures
from conftest import GOLDEN
from sdf_amd import core

# models whose tape goes through libm (see tests/test_gpu.py TRIG)
TRIG = {'ex_gearlike', 'ex_weave', 'ex_knurling'}

FULL = ['c2_example_s27', 'c5_blobby_s30', 'c3_gearlike_s30', 'weave_s24', 'knurling_s27', 'pawn_s27']


def _load(tag):
    path = os.path.join(GOLDEN, 'full_%s.npz' % tag)
    if not os.path.exists(path):
        pytest.skip('%s not generated (tools/make_golden_full.py)' % os.path.basename(path))
    d = np.load(path)
    bounds = tuple(map(tuple, d['bounds']))
    X, Y, Z, _ = core.grid_axes(bounds, d['step'].tolist())
    return d, str(d['fixture']), bounds, X, Y, Z


@pytest.mark.parametrize('tag', ['c2_example_s27', 'c5_blobby_s30', 'pawn_s27'])
def test_oracle_reproduces_reference_soup_at_full_size(tag, ns, oracle_lib):
    d, name, bounds, X, Y, Z = _load(tag)
    f = fixtures.build(name, ns)
    o = oracle_lib.generate(f, X, Y, Z, 32, True)
    assert np.array_equal(o.kinds, d['kinds'])
    assert len(o.points) == 3 * int(d['ntri'])
    assert hashlib.sha256(o.points.tobytes()).digest() == d['sha256'].tobytes()


def test_oracle_reproduces_reference_batches_of_c4_weave_at_2_33(ns, oracle_lib):
    """CPU: the oracle against the reference's full-size run of BASELINE config 4 (full_c4_weave_s33.npz): the skip test on
    a strided sample of ALL batches; classification, triangle count and the reference's sampled triangle for the batches
    that own every 5th of its 5 409 sampled triangles"""
    ref = np.load(os.path.join(GOLDEN, 'full_c4_weave_s33.npz'))
    f = fixtures.build('ex_weave', ns)
    bounds = tuple(map(tuple, ref['bounds']))
    X, Y, Z, _ = core.grid_axes(bounds, ref['step'].tolist())
    assert (len(X), len(Y), len(Z)) == tuple(ref['shape']) == (4097, 4097, 512)
    assert np.array_equal(ref['bounds'], np.load(os.path.join(GOLDEN, 'bounds.npz'))['ex_weave'])
    kinds, tris = ref['kinds'], ref['tris'].astype(np.int64)
    assert len(kinds) == 266256 and int(ref['ntri']) == int(tris.sum()) == 53943912
    assert ((kinds == 2) == (tris > 0)).all() and (kinds != 0).sum() == 37872
    extent = np.ptp(np.array(bounds), axis=0).max()
    offs = np.concatenate([[0], np.cumsum(tris)])
    stride = int(ref['sample_stride'])
    which = np.arange(0, len(ref['sample_tris']), 5)
    owners = np.searchsorted(offs, which * stride, side='right') - 1
    from concurrent.futures import ThreadPoolExecutor

    def oracle_batch(bi):
        return oracle_lib.generate(f, X, Y, Z, 32, True, batch_range=(int(bi), int(bi) + 1))
    n_equal = n_coord = 0
    with ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2)))) as pool:
        for k, bi, o in zip(which, owners, pool.map(oracle_batch, owners)):
            assert o.kinds[bi] == kinds[bi] == 2, bi
            mine = o.points.reshape(-1, 3, 3)
            assert len(mine) == tris[bi], bi
            t = mine[k * stride - offs[bi]]
            assert np.abs(t - ref['sample_tris'][k]).max() <= 1e-5 * extent, bi
            n_equal += int((t == ref['sample_tris'][k]).sum()); n_coord += 9
        some = np.arange(0, 266256, 733)
        for bi, o in zip(some, pool.map(oracle_batch, some)):
            assert o.kinds[bi] == kinds[bi], bi
            assert len(o.points) == 3 * tris[bi], bi
    assert n_equal > 0.999 * n_coord


def _close(a, b, extent):
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= 1e-5 * extent              # north-star tolerance
    assert (a == b).mean() > 0.999


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize('tag', FULL)
def test_full_size_soup_matches_oracle_and_reference(tag, ns, oracle_lib, eng):
    d, name, bounds, X, Y, Z = _load(tag)
    f = fixtures.build(name, ns)
    extent = np.ptp(np.array(bounds), axis=0).max()
    stride = int(d['sample_stride'])
    # passes on (the default)
    m = eng.generate(f, X, Y, Z, 32, True)
    pts, kinds, st, offs = m.points(), m.kinds(), m.stats(), m.batch_offsets()
    m.close()
    assert np.array_equal(kinds, d['kinds'])                 # reference classification, batch by batch
    assert st['triangles'] == int(d['ntri']) and len(pts) == 3 * int(d['ntri'])
    assert offs[0] == 0 and offs[-1] == st['triangles'] and (np.diff(offs) >= 0).all()
    assert np.array_equal(np.diff(offs) > 0, kinds == 2)
    # the reference's own triangles (every 997th)
    sample = pts.reshape(-1, 3, 3)[::stride]
    if name in TRIG:
        _close(sample, d['sample_tris'], extent)
    else:
        assert np.array_equal(sample, d['sample_tris'])
        assert hashlib.sha256(pts.tobytes()).digest() == d['sha256'].tobytes()    # == the reference's soup
    # the oracle's full soup
    o = oracle_lib.generate(f, X, Y, Z, 32, True)
    assert np.array_equal(kinds, o.kinds) and st['n_eval_voxels'] == o.n_eval
    if name in TRIG:
        _close(pts, o.points, extent)
    else:
        assert np.array_equal(pts, o.points)
    del o
    # interval passes off: same device, same libm -> bit for bit, whatever the model
    eng.set_prune(False); eng.set_cull(False)
    try:
        m = eng.generate(f, X, Y, Z, 32, True)
        p0, k0, s0 = m.points(), m.kinds(), m.stats()
        m.close()
    finally:
        eng.set_prune(True); eng.set_cull(True)
    assert s0['n_pruned_instrs'] == 0 and s0['n_sampled_voxels'] == s0['n_eval_voxels']
    assert np.array_equal(k0, kinds) and np.array_equal(p0, pts)


def _sha_of_device_soup(out, T):
    """sha256 of the first T triangles of a flat float64 device tensor, copied out in pieces"""
    h = hashlib.sha256()
    step = 1 << 22
    for t0 in range(0, T, step):
        h.update(out[9 * t0:9 * min(T, t0 + step)].cpu().numpy().tobytes())
    return h.digest()


def _emulated_exchange(eng, f, X, Y, Z, n, chunks, T, nwork):
    """The multi-GPU exchange of sdf_amd/dist.py on ONE device: n ranks x `chunks` shards each mesh their piece of the
    work list into a slab of one buffer -- exactly what the all-gather assembles on every rank, slab (rank r, shard j) at
    position r * chunks + j -- with first-call capacities that may be too small (flagged in the headers, repeated with
    the need the headers report, like collect_sharded does), then sdf_expand_slabs.  Returns (sha256 of the expanded
    float64 soup, gathered headers)."""
    import torch
    S = n * chunks
    cap_items = -(-nwork // S) + 1
    cap_tris = T // S + T // (4 * S) + 4096
    for attempt in range(3):
        sb = eng.slab_bytes(cap_items, cap_tris)
        buf = torch.zeros(S * sb, dtype=torch.uint8, device='cuda:0')
        out = torch.full((9 * (T + 5),), -7.0, dtype=torch.float64, device='cuda:0')
        torch.cuda.synchronize()
        meshes = [eng.generate_compact(f, X, Y, Z, 32, True, (i, S), buf.data_ptr() + i * sb, cap_items, cap_tris) for i in range(S)]
        eng.expand_slabs([buf.data_ptr() + i * sb for i in range(S)], cap_items, cap_tris, out.data_ptr(), T + 5)
        eng.synchronize()
        heads = buf.view(S, sb)[:, :128].cpu().numpy().view(np.int64).reshape(S, 16)
        for m in meshes:
            m.close()
        assert not (heads[:, 2] & 2).any()                                   # no look-back timeout
        assert int(heads[:, 0].sum()) == T and int(heads[:, 1].sum()) == nwork and (heads[:, 9] == nwork).all()
        if not heads[:, 2].any():
            break
        assert attempt < 2
        cap_tris = int(heads[:, 0].max()) + 1024                             # what the headers say is needed
        del buf, out
    assert (out[9 * T:] == -7.0).all().item()
    return _sha_of_device_soup(out, T), heads


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize('tag,n,chunks,passes', [('c5_blobby_s30', 4, 1, True), ('c5_blobby_s30', 4, 2, False), ('c2_example_s27', 8, 1, True),
                                                 ('c2_example_s27', 8, 2, False), ('c3_gearlike_s30', 2, 2, True)],
                         ids=['c5-4ranks', 'c5-4ranks-2chunks-passes-off', 'c2-8ranks', 'c2-8ranks-2chunks-passes-off', 'c3-2ranks-2chunks'])
def test_full_size_exchange_emulated_ranks(tag, n, chunks, passes, ns, eng):
    """BASELINE configs in their multi-GPU shape (configs[4]: blobby 1024^3 on 4 GPUs; the headline example on 8) on one
    device: the soup k_expand assembles from the ranks' slabs is the reference's soup (sha256 of the golden run; the
    libm model: the single-GPU soup of the same device), with the interval passes on and off, one and two shards per rank"""
    d, name, bounds, X, Y, Z = _load(tag)
    f = fixtures.build(name, ns)
    T = int(d['ntri'])
    nwork = int((d['kinds'] != 0).sum())
    want = d['sha256'].tobytes()
    if name in TRIG:                      # (device libm: pinned against the single-GPU soup of this device)
        m = eng.generate(f, X, Y, Z, 32, True)
        want = hashlib.sha256(m.points().tobytes()).digest()
        assert m.n_triangles == T
        m.close()
    eng.set_prune(passes); eng.set_cull(passes)
    try:
        sha, heads = _emulated_exchange(eng, f, X, Y, Z, n, chunks, T, nwork)
    finally:
        eng.set_prune(True); eng.set_cull(True)
    assert sha == want
    assert (int(heads[:, 3].sum()), int(heads[:, 4].sum())) == (int((d['kinds'] == 1).sum()), int((d['kinds'] == 2).sum()))


@pytest.mark.gpu
@pytest.mark.timeout(1800)
def test_c4_weave_at_2_33_sampled_batches_match_oracle(ns, oracle_lib, eng):
    """BASELINE config 4 at its real size on one GPU: the batch classification of every sampled batch and
    the triangles of >= 5 % of the surviving batches (first, last, a seeded random draw from the work list) equal
    the oracle's; the interval passes (79 % of the instructions pruned here) on and off give the same soup"""
    f = fixtures.build('ex_weave', ns)
    b = np.load(os.path.join(GOLDEN, 'bounds.npz'))['ex_weave']
    bounds = tuple(map(tuple, b))
    X, Y, Z, _ = core.grid_axes(bounds, samples=2 ** 33)
    assert (len(X), len(Y), len(Z)) == (4097, 4097, 512)
    extent = np.ptp(np.array(bounds), axis=0).max()
    m = eng.generate(f, X, Y, Z, 32, True)
    try:
        st, kinds, offs = m.stats(), m.kinds(), m.batch_offsets()
        # (the grid is built on the bounds the REFERENCE estimates -- tests/golden/bounds.npz, which the device loop
        # reproduces bit for bit: test_device_bounds_match_reference_for_every_fixture -- so the counts are exact)
        assert st['batches'] == 266256 and st['triangles'] == 53943912
        assert offs[-1] == st['triangles']
        surv = np.flatnonzero(kinds != 0)
        assert len(surv) == 37872
        # the REFERENCE at this size: its own per-batch function (`sdf.core._worker`, imported) over all 266,256 batches of
        # `generate`'s job list, reduced in `generate`'s order (tools/make_golden_c4.py): the classification and the triangle
        # count of EVERY batch, and every 9973rd triangle of its soup
        ref = np.load(os.path.join(GOLDEN, 'full_c4_weave_s33.npz'))
        assert np.array_equal(kinds, ref['kinds'])
        assert np.array_equal(np.diff(offs).astype(np.int64), ref['tris'].astype(np.int64))
        assert st['triangles'] == int(ref['ntri'])
        ref_tris = ref['sample_tris']
        at = np.arange(0, st['triangles'], int(ref['sample_stride']))
        assert len(at) == len(ref_tris) > 5000
        mine = np.stack([m.points_range(int(t), 1).reshape(3, 3) for t in at])
        _close(mine, ref_tris, extent)
        # >= 5 % of the surviving batches, drawn by a SEEDED SHUFFLE (a stride can alias the weave's period), + the ends
        # of the list; the oracle meshes them one by one on the host's cores (ctypes releases the GIL)
        rng = np.random.default_rng(20260922)
        pick = np.unique(np.concatenate([surv[:8], surv[-8:], rng.permutation(surv)[:len(surv) // 20 + 1]]))
        assert len(pick) >= 0.05 * len(surv)
        from concurrent.futures import ThreadPoolExecutor

        def oracle_batch(bi):
            return oracle_lib.generate(f, X, Y, Z, 32, True, batch_range=(int(bi), int(bi) + 1))
        n_equal = n_coord = 0
        with ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 2) - 1))) as pool:
            for bi, o in zip(pick, pool.map(oracle_batch, pick)):
                assert o.kinds[bi] == kinds[bi], bi
                got = m.points_range(offs[bi], offs[bi + 1] - offs[bi])
                assert got.shape == o.points.shape, bi
                if len(got):
                    assert np.abs(got - o.points).max() <= 1e-5 * extent, bi
                    n_equal += int((got == o.points).sum()); n_coord += got.size
        assert n_coord > 2e7 and n_equal > 0.999 * n_coord
        # skipped batches in between: the oracle's skip test agrees on a strided sample of ALL batches
        for bi in range(0, 266256, 2663):
            o = oracle_lib.generate(f, X, Y, Z, 32, True, batch_range=(bi, bi + 1))
            assert o.kinds[bi] == kinds[bi], bi
        # passes off: identical soup (compared by hash on the host, 3.9 GB each)
        h_on = hashlib.sha256()
        step = 1 << 22
        for t0 in range(0, st['triangles'], step):
            h_on.update(m.points_range(t0, min(step, st['triangles'] - t0)).tobytes())
    finally:
        m.close()
    eng.set_prune(False); eng.set_cull(False)
    try:
        m = eng.generate(f, X, Y, Z, 32, True)
        try:
            assert m.stats()['triangles'] == st['triangles'] and np.array_equal(m.kinds(), kinds)
            h_off = hashlib.sha256()
            for t0 in range(0, st['triangles'], step):
                h_off.update(m.points_range(t0, min(step, st['triangles'] - t0)).tobytes())
        finally:
            m.close()
    finally:
        eng.set_prune(True); eng.set_cull(True)
    assert h_on.digest() == h_off.digest()
    # the other meshing scheme (a 244-instruction tape takes the two-pass scheme by default: force one pass, whose
    # sixteen-deep park FIFO this size exercises like no other): the same soup
    eng.set_twopass(0)
    try:
        m = eng.generate(f, X, Y, Z, 32, True)
        try:
            assert m.stats()['triangles'] == st['triangles'] and np.array_equal(m.kinds(), kinds)
            assert np.array_equal(m.batch_offsets(), offs)
            h_one = hashlib.sha256()
            for t0 in range(0, st['triangles'], step):
                h_one.update(m.points_range(t0, min(step, st['triangles'] - t0)).tobytes())
        finally:
            m.close()
    finally:
        eng.set_twopass(-1)
    assert h_one.digest() == h_on.digest()
    # BASELINE configs[3] in its multi-GPU shape ("sharded across 8 x MI355X with RCCL triangle all-gather") on one
    # device: every emulated rank runs the two-pass scheme with the 4-slot libm kernel and k_emit2 writes the compact
    # float32 form into the rank's slab; the soup k_expand assembles from the 8 (16: two shards per rank, interval
    # passes off) slabs is the single-GPU soup, bit for bit
    T, nwork = st['triangles'], len(surv)
    sha8, _ = _emulated_exchange(eng, f, X, Y, Z, 8, 1, T, nwork)
    assert sha8 == h_on.digest()
    eng.set_prune(False); eng.set_cull(False)
    try:
        sha16, _ = _emulated_exchange(eng, f, X, Y, Z, 8, 2, T, nwork)
    finally:
        eng.set_prune(True); eng.set_cull(True)
    assert sha16 == h_on.digest()
