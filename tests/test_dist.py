"""The N>1 path on CPU: world_size-2 gloo processes run sdf_amd.dist.generate_sharded with an
engine whose compute is the CPU oracle (test infrastructure), and must reproduce the
single-process soup, order included."""
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


def _worker(rank, world, port, q):
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
        pts, st = dist.generate_sharded(OracleEngine(), f, X, Y, Z, 32, True)
        import hashlib
        q.put((rank, hashlib.sha256(pts.tobytes()).hexdigest(), len(pts) // 3,
               st['skipped'], st['empty'], st['nonempty'], st['per_rank_triangles']))
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_generate_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
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
