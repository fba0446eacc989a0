"""diagnostics (r06u): does the time per step of gearlike 2^30 with four calls in flight depend on WHICH call slots (lanes) the calls
land on?  N synchronous example calls first (each takes the next slot), then the same 24-step measurement, for N = 0 .. 8."""
import sys, time, gc
import numpy as np
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from sdf_amd import engine, core

eng = engine.get_engine(0)
dev = torch.device('cuda:0')


def run(model, log2, steps, depth, bufs):
    f, _ = bench.build_model(model)
    tape = eng.tape_for(f)
    X, Y, Z, _ = core.grid_axes(core._estimate_bounds(f), samples=2 ** log2)
    inflight = []
    n = [0]

    def step():
        while len(inflight) >= depth:
            m = inflight.pop(0); m.wait(); m.close()
        b = bufs[n[0] % depth]; n[0] += 1
        inflight.append(eng.generate(tape, X, Y, Z, 32, True, out_ptr=b.data_ptr(), out_cap=b.numel() // 9, wait=False))

    def sync():
        while inflight:
            m = inflight.pop(0); m.wait(); m.close()
        eng.synchronize()
    for _ in range(9): step()
    sync()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    return 1e3 * dt / steps


bufs = [torch.empty(9 * 11000000, dtype=torch.float64, device=dev) for _ in range(4)]
fe, _ = bench.build_model('example')
te = eng.tape_for(fe)
A = np.arange(-1.1, 1.1, 2.2 / 128)
for nsync in [0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 3]:
    for _ in range(nsync):
        m = eng.generate(te, A, A, A, 32, True); m.close()
    a = run('gearlike', 30, 24, 4, bufs)
    b = run('gearlike', 30, 24, 1, bufs)
    print('sync example calls before: %d   gearlike 2^30: 4 in flight %.4f ms/step, 1 in flight %.4f' % (nsync, a, b), flush=True)
