"""The N>1 path on CPU: world_size-2 / -3 gloo processes run the exchange protocol of sdf_amd.dist (fixed-capacity
slabs whose headers carry the counts, ONE all-gather per shard, capacity hints, overflow -> every rank retries,
optional chunking for overlap) with an engine whose compute is the CPU oracle (test infrastructure), and must
reproduce the single-process soup, order included.  The device side of the protocol (compact float32 slabs,
k_expand) is covered on one GPU by tests/test_gpu.py::test_slab_exchange_emulated_ranks."""
import os
import sys

import numpy as np
import pytest

import fixtures
from conftest import ROOT, GOLDEN


class OracleMesh:
    def __init__(self, res, lo, hi, nwork):
        self._r = res
        self._lo, self._hi, self._nwork = lo, hi, nwork

    @property
    def n_triangles(self):
        return len(self._r.points) // 3

    def points(self):
        return self._r.points

    def stats(self):
        k = self._r.kinds
        return dict(skipped=int((k == 0).sum()), empty=int((k == 1).sum()), nonempty=int((k == 2).sum()),
                    n_eval_voxels=int(self._r.n_eval), n_ambiguous_cells=int(self._r.n_ambiguous),
                    batches=len(k), n_batches=len(k))

    def close(self):
        pass


class OracleEngine:
    """same `generate(..., shard=)` contract as sdf_amd.engine.Engine, computed by the oracle"""

    def generate(self, sdf, X, Y, Z, batch_size=32, sparse=True, shard=(0, 1)):
        import oracle
        from sdf_amd import dist
        full = oracle.generate(sdf, X, Y, Z, batch_size, sparse)
        work = np.flatnonzero(full.kinds != 0)                  # surviving batches, reference order
        lo, hi = dist.shard_bounds(len(work), shard[0], shard[1])
        # the oracle meshes a contiguous batch range; a shard is a contiguous range of the
        # WORK list, which is a contiguous batch range with the skipped ones inside
        if hi > lo:
            b0, b1 = int(work[lo]), int(work[hi - 1]) + 1
            part = oracle.generate(sdf, X, Y, Z, batch_size, sparse, batch_range=(b0, b1))
            part.kinds = np.where(np.arange(len(part.kinds)) < b0, 3, part.kinds)
        else:
            part = oracle.generate(sdf, X, Y, Z, batch_size, sparse, batch_range=(0, 0))
        # kinds outside the shard: skipped stay 0 (every rank runs the whole prepass)
        k = np.full(len(full.kinds), 3, np.uint8)
        k[full.kinds == 0] = 0
        if hi > lo:
            sel = work[lo:hi]
            k[sel] = full.kinds[sel]
        part.kinds = k
        return OracleMesh(part, lo, hi, len(work))


def _worker(rank, world, port, q, chunks=1):
    import torch.distributed as td
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sdf_amd
        from sdf_amd import core, dist
        ns = {k: getattr(sdf_amd, k) for k in dir(sdf_amd) if not k.startswith('_')}
        f = fixtures.build('ex_example', ns)
        d = np.load(os.path.join(GOLDEN, 'gen_example_s22.npz'))
        X, Y, Z, _ = core.grid_axes(tuple(map(tuple, d['bounds'])), d['step'].tolist())
        assert dist.world_size() == world and dist.rank() == rank
        import hashlib
        eng = OracleEngine()
        soup, st = dist.generate_sharded_device(eng, f, X, Y, Z, 32, True, chunks=chunks)      # first call: upper-bound capacities
        pts = soup.numpy().reshape(-1, 3)
        sha = hashlib.sha256(pts.tobytes()).hexdigest()
        assert st['chunks'] == chunks and st['n_retries'] == 0
        soup2, st2 = dist.generate_sharded_device(eng, f, X, Y, Z, 32, True, chunks=chunks)    # second: capacities from the first
        assert st2['n_retries'] == 0 and st2['slab_bytes'] < st['slab_bytes']
        assert hashlib.sha256(soup2.numpy().tobytes()).hexdigest() == sha
        hints = dist._hints_for(f)
        for k in list(hints):                                                                   # slabs far too small: flagged, repeated
            hints[k] = (1, 7, 3)
        soup3, st3 = dist.generate_sharded_device(eng, f, X, Y, Z, 32, True, chunks=chunks)
        assert st3['n_retries'] >= 1 and st3['triangles'] == st['triangles']
        assert hashlib.sha256(soup3.numpy().tobytes()).hexdigest() == sha
        # two steps in flight on lanes of their own (what bench.py does for N > 1), collected in order
        a = dist.submit_sharded(eng, f, X, Y, Z, 32, True, chunks=chunks, lane=0)
        b = dist.submit_sharded(eng, f, X, Y, Z, 32, True, chunks=chunks, lane=1)
        for step in (a, b):
            soup5, st5 = dist.collect_sharded(step)
            assert hashlib.sha256(soup5.numpy().tobytes()).hexdigest() == sha and st5['n_retries'] == 0
        pts4, st4 = dist.generate_sharded(eng, f, X, Y, Z, 32, True)                            # the ndarray front
        assert hashlib.sha256(pts4.tobytes()).hexdigest() == sha
        q.put((rank, sha, len(pts) // 3,
               st['skipped'], st['empty'], st['nonempty'], st['per_rank_triangles']))
    except BaseException:
        import traceback
        q.put((rank, 'ERROR', traceback.format_exc()))
        raise
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('world,chunks', [(2, 1), (3, 1), (2, 2)])
def test_sharded_generate_over_gloo(world, chunks):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world + 7 * chunks
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    errors = [o for o in out if o[1] == 'ERROR']
    if errors:
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.terminate()
        pytest.fail('rank %d failed:\n%s' % (errors[0][0], errors[0][2]))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    d = np.load(os.path.join(GOLDEN, 'gen_example_s22.npz'))
    want = bytes(d['sha256']).hex()
    for rank, sha, ntri, sk, em, ne, per in out:
        assert sha == want, 'rank %d: all-gathered soup differs from the reference' % rank
        assert ntri == int(d['ntri']) == sum(per)
        assert (sk, em, ne) == (44, 60, 112)              # BASELINE config 1 classification
    assert all(o[6] == out[0][6] for o in out) and min(out[0][6]) > 0


def _agree_worker(rank, world, port, q, missing_on):
    """dist._native_comm up to (not including) the RCCL communicator: the ranks' agreement about librccl"""
    import torch.distributed as td
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    td.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from sdf_amd import dist, engine

        class FakeEngine:
            lib = object()

        class StubComm:
            created = []

            @staticmethod
            def available(lib):
                return (False, 'no librccl here') if rank in missing_on else (True, '')

            @staticmethod
            def unique_ids(lib, n_lanes=2):
                return bytes([7]) * (engine.COMM_ID_BYTES * n_lanes)

            def __init__(self, eng, ids, r, w):
                StubComm.created.append((ids, r, w))

            def close(self):
                pass

        real = engine.Comm
        engine.Comm = StubComm
        try:
            try:
                fake = FakeEngine()
                comm = dist._native_comm(fake, td, None)
                assert dist._native_comm(fake, td, None) is comm and len(StubComm.created) == 1      # (created once per engine and group)
                got = ('comm', StubComm.created[0][0][:4], StubComm.created[0][1:])
            except RuntimeError as e:
                got = ('refused', str(e))
        finally:
            engine.Comm = real
            dist._COMMS.clear()
        td.barrier()                       # (nobody is left behind in a collective)
        q.put((rank, got))
    except BaseException:
        import traceback
        q.put((rank, ('ERROR', traceback.format_exc())))
        raise
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('missing_on', [(), (1,), (0,)])
def test_native_communicator_is_agreed_on_by_all_ranks(missing_on):
    """a rank whose process cannot load librccl must not leave the others in a collective: every rank refuses the native
    exchange alike (and takes the torch.distributed path) -- or every rank creates its communicator from rank 0's ids"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 2
    port = 31700 + (os.getpid() % 2000) + 3 * len(missing_on) + sum(missing_on)
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q, missing_on)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert out[r][0] != 'ERROR', out[r][1]
        if missing_on:
            assert out[r][0] == 'refused' and 'rank %d: no librccl here' % missing_on[0] in out[r][1]
        else:
            assert out[r][0] == 'comm' and out[r][1] == bytes([7]) * 4 and out[r][2] == (r, world)
