#!/usr/bin/env python3
"""Soundness audit of the interval interpreter on the CPU (no GPU):   python tools/interval_audit.py [trees per kind]
Random CSG / array / composed-leaf trees (the generators of tests/test_gpu.py), 600 boxes each; the CPU checker's
values at 16 points of every box must lie inside the interval that the host build of ia_run_tape computes for the
box (tests/test_interval_host.py runs a small fixed sample of this).  Prints any violation."""
import sys, os, numpy as np, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sdf_amd, oracle, fixtures
import test_interval_host as tih
import test_gpu as tg
ns = {k: getattr(sdf_amd, k) for k in dir(sdf_amd) if not k.startswith('_')}
so = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'libia_tape_audit.so')
subprocess.check_call([tih.HIPCC, '--offload-host-only', '-O1', '-std=c++17', '-ffp-contract=off', '-w', '-fPIC', '-shared', '-I', os.path.join(ROOT, 'sdf_amd', 'csrc'), '-o', so, os.path.join(ROOT, 'tests', 'native', 'interval_tape_host.hip')])
lib = ctypes.CDLL(so); lib.ia_tape_boxes.restype = ctypes.c_int
lib.ia_tape_boxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
P = np.load(os.path.join(ROOT, 'tests', 'golden', 'values.npz'))['P']
bad = 0; n = 0
for kind, gen in (('csg', tg._random_csg), ('array', tg._random_array_tree), ('leaf', tg._random_leaf_tree)):
    for seed in range(100, 100 + (int(sys.argv[1]) if len(sys.argv) > 1 else 50)):
        rng = np.random.default_rng(90000 + seed * 3 + len(kind))
        try:
            f = gen(rng, ns)
            tih._check_tape_enclosure('%s-%d' % (kind, seed), f, lib, oracle, P, seed, nb=600)
            n += 1
        except AssertionError as e:
            bad += 1; print('VIOLATION', kind, seed, str(e)[:300], flush=True)
        except Exception as e:
            print('skip', kind, seed, type(e).__name__, str(e)[:100])
print('trees checked', n, 'violations', bad)
