"""Time of `_estimate_bounds` on the device per model (GPU box only, diagnostics): python tools/boundstime.py
(the variants that were measured: csrc/sdf_bounds.hip)"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import sdf_amd, fixtures
from sdf_amd import engine
ns = {k: getattr(sdf_amd, k) for k in dir(sdf_amd) if not k.startswith('_')}
eng = engine.get_engine(0)
for name in ('ex_example', 'ex_gearlike', 'ex_blobby', 'ex_weave', 'ex_knurling', 'ex_pawn'):
    f = fixtures.build(name, ns)
    t = eng.tape_for(f)
    for _ in range(3): b = eng.estimate_bounds(t)
    t0 = time.perf_counter()
    for _ in range(10): b = eng.estimate_bounds(t)
    print('%-12s %.3f ms per estimate_bounds' % (name, 1e2 * (time.perf_counter() - t0)), flush=True)
