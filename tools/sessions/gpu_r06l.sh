#!/bin/bash
# r06l: host expansion with non-temporal stores: workers sweep + trace
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06l}
mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/rec.py <<'PY'
import sys, time, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench
from sdf_amd import engine, core
w=int(sys.argv[1])
eng=engine.get_engine(0)
f,_=bench.build_model('example')
b=core._estimate_bounds(f)
X,Y,Z,_=core.grid_axes(b,None,2**27)
t=eng.tape_for(f)
m=eng.generate(t,X,Y,Z,32,True); want=m.points().copy(); m.close()
ts=[]
for i in range(10):
    m=eng.generate(t,X,Y,Z,32,True,records=True); t2=time.perf_counter(); p=m.points(w); t3=time.perf_counter(); m.close()
    if i==3: assert np.array_equal(p.view(np.uint64), want.view(np.uint64))
    del p
    ts.append(1e3*(t3-t2))
print('workers',w,'pieces',os.environ.get('SDF_REC_PIECES'),'block',os.environ.get('SDF_REC_BLOCK'),'points ms',round(np.median(ts[2:]),3), [round(x,2) for x in ts])
PY
for w in 8 32; do SDF_REC_TRACE=1 python /tmp/rec.py $w 2>&1 | tail -3; done > $O/trace.txt 2>&1
cat $O/trace.txt
