import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import sdf_amd as s
from sdf_amd import core, engine
import fixtures
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
eng = engine.get_engine(0)
for name, samples in (('ex_example', 2**27), ('ex_gearlike', 2**27), ('ex_blobby', 2**27)):
    f = fixtures.build(name, ns)
    X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), None, samples)
    for cull in (1, 0):
        eng.set_cull(cull)
        for _ in range(3):
            m = eng.generate(f, X, Y, Z, 32, True); st = m.stats(); m.close()
        print(name, 'cull', cull, 'sampled %.1f%%' % (100.0 * st['n_sampled_voxels'] / st['n_eval_voxels']), 'mesh ms %.3f' % st['ms_mesh'], 'pruned %.1f%%' % (100.0*st['n_pruned_instrs']/max(st['n_batch_instrs'],1)))
