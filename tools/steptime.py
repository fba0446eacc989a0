import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import torch
torch.cuda.set_device(0)
import bench
from sdf_amd import core, engine
eng = engine.get_engine(0)
f,_ = bench.build_model('example')
tape = eng.tape_for(f)
X,Y,Z,_ = core.grid_axes(bench.EXAMPLE_BOUNDS, samples=2**27)
buf = torch.empty(9*(1<<22), dtype=torch.float64, device='cuda:0')
T = {}
def tick(k, t0):
    T[k] = T.get(k,0)+time.perf_counter()-t0
N=200
for i in range(N+5):
    if i==5: T.clear(); t_all=time.perf_counter()
    t0=time.perf_counter(); m = eng.generate(tape, X, Y, Z, 32, True, out_ptr=buf.data_ptr(), out_cap=buf.numel()//9); tick('generate',t0)
    t0=time.perf_counter(); t = m.n_triangles; tick('ntri',t0)
    t0=time.perf_counter(); st = m.stats(); tick('stats',t0)
    t0=time.perf_counter(); m.close(); tick('close',t0)
tot=time.perf_counter()-t_all
print('total ms/step', 1e3*tot/N, {k: round(1e3*v/N,4) for k,v in T.items()}, 'device: prepass %.3f mesh %.3f emit %.3f total %.3f'%(st['ms_prepass'],st['ms_mesh'],st['ms_emit'],st['ms_total']))
