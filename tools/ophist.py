#!/usr/bin/env python3
"""Which tape instructions survive the interval pruning, per surviving batch?  (GPU box only, diagnostics)

    python tools/ophist.py name:log2samples ...        e.g. weave:33 gearlike:30

Per model: mean number of surviving instructions per surviving batch, by opcode -- the instruction mix the
interpreter actually runs."""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np   # noqa: E402
import sdf_amd as s  # noqa: E402
from sdf_amd import core, engine, tape  # noqa: E402
import fixtures      # noqa: E402

ns = {k: getattr(s, k) for k in dir(s) if not k.startswith('_')}
eng = engine.get_engine(0)
for job in sys.argv[1:] or ['weave:27']:
    name, k = job.split(':')
    f = fixtures.build('ex_' + name, ns)
    t = tape.lower(f)
    X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), None, 2 ** int(k))
    m = eng.generate(f, X, Y, Z, 32, True)
    st = m.stats()
    masks = m.prune_masks()[np.isin(m.kinds(), (1, 2))]
    m.close()
    n = t.n_instr - 1
    bits = ((masks[:, :8, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(len(masks), 256)[:, :n]
    keep = 1.0 - bits.mean(axis=0)
    ops = [l.split('post=')[0].split(';')[-1].split()[-1] if 'post=' in l else l.split()[-1] for l in t.disassemble().split('\n')[:n]]
    per = defaultdict(float)
    tot = defaultdict(int)
    for i in range(n):
        per[ops[i]] += keep[i]
        tot[ops[i]] += 1
    print('%s 2^%s: %d instructions, %d surviving batches, %.1f instructions survive per batch (pruned %.1f%%); sampled %.1f%%'
          % (name, k, n, len(masks), keep.sum(), 100.0 * st['n_pruned_instrs'] / max(st['n_batch_instrs'], 1),
             100.0 * st['n_sampled_voxels'] / max(st['n_eval_voxels'], 1)))
    for op in sorted(per, key=lambda o: -per[o]):
        print('    %-16s %6.2f of %3d per batch' % (op, per[op], tot[op]))
    lens = (1 - bits).sum(axis=1)
    print('    surviving tape length per batch: min %d  p10 %d  median %d  p90 %d  max %d' % tuple(np.percentile(lens, [0, 10, 50, 90, 100]).astype(int)))
