#!/usr/bin/env python3
"""What does the interval prepass (csrc/sdf_prune.h) remove?  (GPU box only, diagnostics)

    python tools/prunestat.py [model ...]      # models: the names of tests/fixtures.py, default a few examples

Per model: instructions per tape, share of (batch, instruction) pairs pruned, per-instruction skip
rate, and the meshing-kernel time with the prepass on and off.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np   # noqa: E402
import torch         # noqa: E402,F401  (first: see INTEGRATION.md section 4)
import sdf_amd as s  # noqa: E402
from sdf_amd import core, engine, tape  # noqa: E402
import fixtures      # noqa: E402


def main():
    names = sys.argv[1:] or ['ex_example', 'ex_gearlike', 'ex_blobby', 'ex_knurling']
    ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
    eng = engine.get_engine(0)
    for name in names:
        f = fixtures.build(name, ns)
        t = tape.lower(f)
        X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), None, 2 ** 24)
        res = {}
        for on in (True, False):
            eng.set_prune(on)
            for _ in range(3):
                m = eng.generate(f, X, Y, Z, 32, True)
                st = m.stats()
                masks = m.prune_masks()[np.isin(m.kinds(), (1, 2))] if on else None
                m.close()
            res[on] = (st, masks)
        eng.set_prune(True)
        st, masks = res[True]
        n = t.n_instr - 1
        print('%s: %d instructions, %d batches sampled, pruned %.1f%%, k_mesh %.3f ms (prepass %.3f) vs %.3f ms (%.3f) without'
              % (name, n, len(masks), 100.0 * st['n_pruned_instrs'] / max(st['n_batch_instrs'], 1), st['ms_mesh'],
                 st['ms_prepass'], res[False][0]['ms_mesh'], res[False][0]['ms_prepass']))
        bits = (masks[:, :8, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1
        rate = bits.reshape(len(masks), 256)[:, :n].mean(axis=0)
        forced = ((masks[:, 8:, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(len(masks), 256)[:, :n].mean(axis=0)
        names_ = t.disassemble().split('\n')
        for i in range(n):
            print('   %3d %-60s skip %5.1f%%  forced %5.1f%%' % (i, names_[i][:60], 100 * rate[i], 100 * forced[i]))


if __name__ == '__main__':
    main()
