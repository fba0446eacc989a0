#!/bin/bash
# r06j: records mode (sdf_generate_records + host-thread expansion): its tests, the drop-in tests around it, and the bench line's generate_e2e
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06j}
mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt; lscpu | head -25 >> $O/nproc.txt
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "records or drop_in or recycled_pinned or stl_records or weld or generate_to_device" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-other-configs --no-f32-envelope > $O/bench.txt 2> $O/bench.err
python - "$O" <<'PY'
import json,sys
for l in open(sys.argv[1]+'/bench.txt'):
    if l.startswith('{"metric"'):
        r=json.loads(l)
        print('ms/step', r['ms_per_step'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'parity', r['parity_check'])
        print('e2e', json.dumps(r['generate_e2e'])[:1200])
        print('sustained', r['sustained']['ms_per_step'])
        print('cpu', json.dumps(r['cpu_baseline'])[:300])
PY
for w in 1 2 4 8 16 32 64; do
python - $w <<'PY'
import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench
from sdf_amd import engine, core
w=int(sys.argv[1])
eng=engine.get_engine(0)
f,_=bench.build_model('example')
b=core._estimate_bounds(f)
X,Y,Z,_=core.grid_axes(b,None,2**27)
t=eng.tape_for(f)
m=eng.generate(t,X,Y,Z,32,True); m.close()
ts=[]; tg=[]
for i in range(8):
    t1=time.perf_counter(); m=eng.generate(t,X,Y,Z,32,True,records=True); t2=time.perf_counter(); p=m.points(w); t3=time.perf_counter(); m.close(); del p
    tg.append(1e3*(t2-t1)); ts.append(1e3*(t3-t2))
print('workers',w,'generate_records ms',round(np.median(tg[2:]),3),'points ms',round(np.median(ts[2:]),3), [round(x,2) for x in ts])
PY
done 2>&1 | tee $O/workers.txt
