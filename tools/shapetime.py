#!/usr/bin/env python3
"""k_mesh time per launch shape (GPU box, tuning):  python tools/shapetime.py
Runs each model in a child process per SDF_MESH_SHAPE (0 = 1024 threads x 1 sample per lane, 1 = 512 x 2,
3 = 1024 x 2) and prints the meshing-kernel time."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = [('ex_example', 2 ** 27), ('ex_gearlike', 2 ** 27), ('ex_blobby', 2 ** 27), ('ex_weave', 2 ** 24), ('ex_knurling', 2 ** 24)]
CHILD = r'''
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests')
import torch
import sdf_amd as s
from sdf_amd import core, engine, tape
import fixtures
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
f = fixtures.build(%(name)r, ns)
t = tape.lower(f)
eng = engine.get_engine(0)
X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), None, %(samples)d)
for _ in range(4):
    m = eng.generate(f, X, Y, Z, 32, True); st = m.stats(); m.close()
print('RESULT', t.pmax if hasattr(t, 'pmax') else -1, st['ms_mesh'], st['ms_prepass'], st['n_triangles'], 100.0 * st['n_sampled_voxels'] / max(st['n_eval_voxels'], 1))
'''
for name, samples in MODELS:
    for shape in ('0', '1', '3'):
        env = dict(os.environ, SDF_MESH_SHAPE=shape)
        p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, name=name, samples=samples)], env=env, capture_output=True, text=True)
        r = [l for l in p.stdout.splitlines() if l.startswith('RESULT')]
        print(name, 'shape', shape, r[-1] if r else ('FAILED ' + p.stderr[-300:]))
