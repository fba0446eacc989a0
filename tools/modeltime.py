"""Device time per model with the interval passes on / off (run on the GPU box):
    python tools/modeltime.py [name:log2samples ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import time
import numpy as np
import sdf_amd as s
from sdf_amd import core, engine
import fixtures
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
eng = engine.get_engine(0)
on_only = '--on-only' in sys.argv
BS = int(os.environ.get('MODELTIME_BS', '32'))
jobs = [a for a in sys.argv[1:] if not a.startswith('--')] or ['example:27', 'gearlike:27', 'gearlike:30', 'blobby:27', 'blobby:30', 'weave:24', 'weave:27', 'knurling:24']
for job in jobs:
    name, k = job.split(':')
    f = fixtures.build('ex_' + name, ns)
    X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), None, 2 ** int(k))
    for on in ((1,) if on_only else (1, 0)):
        eng.set_prune(on); eng.set_cull(on)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            m = eng.generate(f, X, Y, Z, BS, True); st = m.stats(); m.close()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print('%-9s 2^%s %dx%dx%d passes %d: wall %.2f ms, prepass %.3f mesh %.3f ms (kernel %d, %d retries); batches %d work %d tris %d; sampled %.1f%% pruned %.1f%%'
              % (name, k, len(X), len(Y), len(Z), on, 1e3 * best, st['ms_prepass'], st['ms_mesh'], st['mesh_kernel'], st['n_retries'], st['batches'],
                 st['empty'] + st['nonempty'], st['triangles'], 100.0 * st['n_sampled_voxels'] / max(st['n_eval_voxels'], 1),
                 100.0 * st['n_pruned_instrs'] / max(st['n_batch_instrs'], 1)), flush=True)
eng.set_prune(1); eng.set_cull(1)
