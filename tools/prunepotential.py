"""How much shorter would tapes get if they were pruned per 8^3 (4^3) cells instead of per batch?
Runs the per-batch interval prepass with small batch sizes and reports the pruned share over the
surviving batches (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sdf_amd as s
from sdf_amd import core, engine
import fixtures
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
eng = engine.get_engine(0)
eng.set_cull(0)
for job in sys.argv[1:] or ['example:27', 'gearlike:27', 'blobby:27', 'weave:27', 'weave:24']:
    name, k = job.split(':')
    f = fixtures.build('ex_' + name, ns)
    X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), None, 2 ** int(k))
    for bs in (32, 16, 8, 4):
        m = eng.generate(f, X, Y, Z, bs, True); st = m.stats(); m.close()
        print('%-9s 2^%s bs %2d: batches %d surviving %d  pruned %.1f%% of the surviving batches\' instructions'
              % (name, k, bs, st['batches'], st['empty'] + st['nonempty'], 100.0 * st['n_pruned_instrs'] / max(st['n_batch_instrs'], 1)), flush=True)
