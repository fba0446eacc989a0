#!/usr/bin/env python3
"""Dynamic instruction counts per tape instruction (GPU box only, diagnostics).

    python tools/opcount.py [f64|f32]

Runs the model ladder of tools/opcost.py once each (dense 256^3 grid) under
`rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY`
and prints the k_mesh counters per wave-iteration (one tape run over NS*64 samples).
"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.opcost import MODELS  # noqa: E402

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import sdf_amd as s
from sdf_amd import engine, tape
ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
eng = engine.get_engine(0)
eng.precision = engine.PRECISION_F64 if %(prec)r == 'f64' else engine.PRECISION_F32
A = np.arange(-1.2, 1.2, 2.4 / 256)
for name, expr in %(models)r.items():
    f = eval(expr, ns)
    m = eng.generate(f, A, A, A, 32, False)
    st = m.stats(); m.close()
    print('RESULT', name, tape.lower(f).n_instr - 1, st['n_eval_voxels'])
'''


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f64'
    ns = int(os.environ.get('OPCOUNT_NS', '2'))
    out = '/tmp/opcount_%s' % prec
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_WAVE_CYCLES',
           'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_LDS', '--output-format', 'csv', '-d', out, '-o', 'oc', '--',
           sys.executable, '-c', CHILD % dict(root=ROOT, prec=prec, models=MODELS)]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
    res = [ln.split()[1:] for ln in p.stdout.splitlines() if ln.startswith('RESULT')]
    rows = []
    for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if 'k_mesh' in r['Kernel_Name']]
    by = {}
    for r in rows:
        by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    disp = [by[k] for k in sorted(by)]
    if len(disp) != len(res):
        print('dispatch/model mismatch', len(disp), len(res), p.stderr[-500:])
    print('%-14s %5s %9s %9s %8s %10s %9s %9s %9s' % ('model', 'instr', 'VALU/it', 'SALU/it', 'SMEM/it', 'wavecyc/it', 'wait%', 'stall%', 'valu_act%'))
    for (name, n_instr, n_eval), d in zip(res, disp):
        its = int(n_eval) / (64.0 * ns)
        wc = d['SQ_WAVE_CYCLES']
        print('%-14s %5s %9.1f %9.1f %8.1f %10.1f %9.1f %9.1f %9.1f' % (
            name, n_instr, d['SQ_INSTS_VALU'] / its, d['SQ_INSTS_SALU'] / its, d['SQ_INSTS_SMEM'] / its, 4 * wc / its,
            100 * d['SQ_WAIT_ANY'] / wc, 100 * d['SQ_WAIT_INST_ANY'] / wc, 100 * d['SQ_ACTIVE_INST_VALU'] / wc))


if __name__ == '__main__':
    main()
