"""One call at a time, three ways (GPU box only, diagnostics): the soup into library memory, into a caller buffer,
and submitted asynchronously + collected -- wall time of the call / of reading its statistics / of closing it, the
prepass and k_mesh by HIP events.  `python tools/asyncprobe.py`"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import bench
from sdf_amd import core, engine
eng = engine.get_engine(0)
for model, log2 in (('gearlike', 30), ('blobby', 30)):
    f, _ = bench.build_model(model)
    tape = eng.tape_for(f)
    X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), samples=2 ** log2)
    buf = torch.empty(9 * 16_000_000, dtype=torch.float64, device='cuda:0')
    for mode in ('sync-lib', 'sync-buf', 'async-buf'):
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            if mode == 'sync-lib':
                m = eng.generate(tape, X, Y, Z, 32, True)
            elif mode == 'sync-buf':
                m = eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel() // 9)
            else:
                m = eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel() // 9, wait=False)
                m.wait()
            t1 = time.perf_counter()
            st = m.stats()
            t2 = time.perf_counter()
            m.close()
            t3 = time.perf_counter()
            ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), st['ms_prepass'], st['ms_mesh'], m.emitted, st['n_retries']))
        print(model, mode, ['%.3f/%.3f/%.3f pre %.3f mesh %.3f em %s r %d' % t for t in ts[3:6]], flush=True)
