import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import sdf_amd as s
from sdf_amd import core, engine
import fixtures
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
eng = engine.get_engine(0)
for name in ['ex_example', 'sphere']:
    f = fixtures.build(name, ns) if name != 'sphere' else s.sphere(0.8)
    X, Y, Z, _ = core.grid_axes(((-0.8454303741455078,)*3, (0.8454312324523926,)*3), None, 2 ** 27)
    for _ in range(3):
        m = eng.generate(f, X, Y, Z, 32, True); st = m.stats(); m.close()
    print(name, 'tris', st['triangles'], 'ambiguous cells', st['n_ambiguous_cells'], 'work', st['empty'] + st['nonempty'], 'mesh ms %.3f' % st['ms_mesh'], flush=True)
