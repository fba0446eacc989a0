#!/usr/bin/env python3
"""Fixed costs of the multi-GPU exchange path, measured on ONE GPU with a one-rank RCCL group (GPU box):
    python tools/disttime.py [steps [log2 samples]]      (a tiny grid, e.g. 16, shows the host's share of a step)
The same job as bench.py (example at 512^3) through sdf_amd.dist: mesh into a slab, all-gather (one rank: a copy),
k_expand, one host synchronisation per step; one and two steps in flight."""
import gc, os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as td
import bench
from sdf_amd import core, engine, dist

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
torch.cuda.set_device(0)
td.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
eng = engine.get_engine(0)
f, _ = bench.build_model('example')
tape = eng.tape_for(f)
LOG2 = int(sys.argv[2]) if len(sys.argv) > 2 else 27
X, Y, Z, _ = core.grid_axes(bench.EXAMPLE_BOUNDS, samples=2 ** LOG2)
dev = torch.device('cuda', 0)
ONLY = os.environ.get('DISTTIME_ONLY')      # e.g. "1,2": chunks 1 with two steps in flight only; "1,1;1,2": those two
for chunks in (1, 2):
    for depth in (1, 2):
        if ONLY and '%d,%d' % (chunks, depth) not in ONLY.split(';'):
            continue
        # (the same loop twice, the second pass timed: a lane's first step allocates its slabs and its soup, a call slot's
        # first use its 1.2 GB of park slots -- with two steps in flight that is a second slot, 45 ms once)
        for timed in (False, True):
            inflight, acc = [], []
            torch.cuda.synchronize()
            gc.collect(); gc.disable()       # (a full collection is 30 - 50 ms with torch imported: not inside a timed pass)
            if timed and os.environ.get('SDF_POOL_TRACE'):
                print('-- timed pass starts', flush=True)
            t0 = time.perf_counter()
            for i in range(steps if timed else 8):
                while len(inflight) >= depth:
                    acc.append(dist.collect_sharded(inflight.pop(0))[1])
                inflight.append(dist.submit_sharded(eng, tape, X, Y, Z, 32, True, device=dev, chunks=chunks, lane=i % 2))
            while inflight:
                acc.append(dist.collect_sharded(inflight.pop(0))[1])
            torch.cuda.synchronize()
            gc.enable()
        dt = (time.perf_counter() - t0) / steps
        print('chunks %d, %d step(s) in flight: %.3f ms per step; device: mesh %.3f exchange %.3f expand %.3f ms; slab %.1f MB; retries %d'
              % (chunks, depth, 1e3 * dt, np.mean([a['ms_mesh'] for a in acc]), np.mean([a['ms_exchange'] for a in acc]),
                 np.mean([a['ms_expand'] for a in acc]), acc[-1]['slab_bytes'] / 1e6, sum(a.get('n_retries', 0) for a in acc)), flush=True)
td.destroy_process_group()
