#!/usr/bin/env python3
"""Per-op cost probe for the tape interpreter (GPU box only, diagnostics):

    SDF_MESH_PROF=1 python tools/opcost.py [f64|f32]

Meshes a dense 256^3 grid (sparse=False, 512 batches) for a ladder of synthetic models and prints
the sampling-phase cycles per (CU, sample) that libsdf_hip's SDF_MESH_PROF counters report, so
differences between rungs give the cost of one more tape instruction of a kind.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODELS = {
    'sphere': 'sphere(1)',
    'sphere_x2': 'sphere(1) | sphere(0.9, (0.1, 0, 0))',
    'sphere_x4': 'sphere(1) | sphere(0.9, (0.1, 0, 0)) | sphere(0.8, (0, 0.1, 0)) | sphere(0.7, (0, 0, 0.1))',
    'sphere_x8': ' | '.join('sphere(%g, (%g, 0, 0))' % (1 - 0.05 * i, 0.02 * i) for i in range(8)),
    'plane_x8': ' & '.join('plane((%g, %g, 1), (0, 0, %g))' % (0.1 * i, -0.05 * i, 0.5 - 0.01 * i) for i in range(8)),
    'box': 'box(1.5)',
    'box_x4': ' | '.join('box(%g, (%g, 0, 0))' % (1.5 - 0.1 * i, 0.02 * i) for i in range(4)),
    'cylinder_x4': ' | '.join('cylinder(%g)' % (0.5 - 0.05 * i) for i in range(4)),
    'translate_x4': ' | '.join('sphere(%g).translate((%g, 0, 0))' % (1 - 0.05 * i, 0.02 * i) for i in range(4)),
    'rotate_x4': ' | '.join('sphere(%g, (0.1, 0, 0)).rotate(%g, X)' % (1 - 0.05 * i, 0.3 * i + 0.1) for i in range(4)),
    'smooth_x4': 'union(' + ', '.join('sphere(%g, (%g, 0, 0))' % (1 - 0.05 * i, 0.3 * i) for i in range(4)) + ', k=0.2)',
    'example': '(sphere(1) & box(1.5)) - (cylinder(0.5).orient(X) | cylinder(0.5).orient(Y) | cylinder(0.5).orient(Z))',
    'circ_array': 'cylinder(0.1).translate((0.8, 0, 0)).circular_array(12)',
    # the weave model's chain, one construct at a time (each rung adds what the name says)
    'rbox': 'rounded_box([1.2, 0.5, 0.25], 0.1)',
    'rbox_t': 'rounded_box([1.2, 0.5, 0.25], 0.1).translate((0.5, 0, 0.0625))',
    'rbox_tb': 'rounded_box([1.2, 0.5, 0.25], 0.1).translate((0.5, 0, 0.0625)).bend_linear(X * 0.25, X * 0.75, Z * -0.1875, ease.in_out_quad)',
    'rbox_tbc': 'rounded_box([1.2, 0.5, 0.25], 0.1).translate((0.5, 0, 0.0625)).bend_linear(X * 0.25, X * 0.75, Z * -0.1875, ease.in_out_quad).circular_array(3, 0)',
    'rbox_tbcr': 'rounded_box([1.2, 0.5, 0.25], 0.1).translate((0.5, 0, 0.0625)).bend_linear(X * 0.25, X * 0.75, Z * -0.1875, ease.in_out_quad).circular_array(3, 0).repeat((0.9, 1.8, 0), padding=1)',
    'twist': 'box((0.6, 0.6, 1.8)).twist(1.5)',
    'gearlike': """(sphere(2) & slab(z0=-0.5, z1=0.5).k(0.1)) - cylinder(1).k(0.1) - cylinder(0.25).circular_array(16, 2).k(0.1)""",
    'blobby': """union(*[sphere(0.4, (0.6 * ((i * 7) % 5 - 2) / 2, 0.6 * ((i * 3) % 5 - 2) / 2, 0.6 * ((i * 5) % 5 - 2) / 2)) for i in range(7)], k=0.3)""",
}

CHILD = r'''
import sys, os
sys.path.insert(0, %(root)r)
import numpy as np
import sdf_amd as s
from sdf_amd import engine, tape
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
f = eval(%(expr)r, ns)
eng = engine.get_engine(0)
eng.precision = engine.PRECISION_F64 if %(prec)r == 'f64' else engine.PRECISION_F32
A = np.arange(-1.2, 1.2, 2.4 / 256)
for _ in range(3):
    m = eng.generate(f, A, A, A, 32, False)
    st = m.stats(); m.close()
t = tape.lower(f)
print('RESULT', t.n_instr, st['n_eval_voxels'], st['ms_mesh'], st['triangles'])
'''


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f64'
    env = dict(os.environ, SDF_MESH_PROF='1', SDF_CULL='0', SDF_PRUNE='0')     # every sample through the whole tape
    only = [a for a in sys.argv[2:]]
    print('%-14s %6s %10s %10s %12s %12s' % ('model', 'instr', 'ms_mesh', 'tris', 'cyc/sample', 'd(cyc)/instr'))
    base = None
    for name, expr in MODELS.items():
        if only and name not in only and name != 'sphere':
            continue
        p = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, expr=expr, prec=prec)], env=env,
                           capture_output=True, text=True)
        m = re.findall(r'sample (\d+) count (\d+) \(of which placing the parked batch \d+\) list (\d+) emit (\d+)', p.stderr)
        r = re.search(r'RESULT (\d+) (\d+) ([\d.]+) (\d+)', p.stdout)
        if not m or not r:
            print(name, 'FAILED', p.stderr[-400:])
            continue
        sample_cyc = int(m[-1][0])
        n_instr, n_eval, ms, tris = int(r.group(1)) - 1, int(r.group(2)), float(r.group(3)), int(r.group(4))
        cps = sample_cyc / n_eval * 1.0      # WG-cycles per sample (one WG per CU)
        if name == 'sphere':
            base = cps
        d = (cps - base) / (n_instr - 1) if base is not None and n_instr > 1 else float('nan')
        print('%-14s %6d %10.3f %10d %12.3f %12.3f' % (name, n_instr, ms, tris, cps, d))


if __name__ == '__main__':
    main()
