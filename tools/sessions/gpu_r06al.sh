#!/bin/bash
# r06al: the helpers of the host expansion one per core complex of the result block's NUMA node (cpu_plan, sdf_expand_host.h) against threads
# the scheduler places (SDF_REC_AFFINITY=0): workers sweep, alternating, + the bench line's generate_e2e both ways
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06al}
mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/rec.py <<'PY'
import sys, time, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench
from sdf_amd import engine, core
eng=engine.get_engine(0)
f,_=bench.build_model('example')
b=core._estimate_bounds(f)
X,Y,Z,_=core.grid_axes(b,None,2**27)
t=eng.tape_for(f)
m=eng.generate(t,X,Y,Z,32,True); want=m.points().copy(); m.close()
for w in [int(x) for x in sys.argv[1:]]:
    ts=[]
    for i in range(10):
        m=eng.generate(t,X,Y,Z,32,True,records=True); t2=time.perf_counter(); p=m.points(w); t3=time.perf_counter(); m.close()
        if i==3: assert np.array_equal(p.view(np.uint64), want.view(np.uint64))
        del p
        ts.append(1e3*(t3-t2))
    print('affinity',os.environ.get('SDF_REC_AFFINITY','1'),'workers',w,'points ms median',round(np.median(ts[2:]),3),'min',round(min(ts[2:]),3),'max',round(max(ts[2:]),3), flush=True)
PY
for rep in 1 2; do for a in 1 0; do SDF_REC_AFFINITY=$a python /tmp/rec.py 4 8 12 16 24 32 2>&1 | grep affinity; done; done | tee $O/sweep.txt
SDF_REC_TRACE=1 python /tmp/rec.py 16 2>&1 | grep "records\]" | tail -2 | cut -c1-900 | tee $O/trace16.txt
for a in 1 0; do SDF_REC_AFFINITY=$a timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_aff$a.txt 2>&1; done
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l); e=r['generate_e2e']
            print(os.path.basename(f), 'e2e', e['wall_ms'], 'rows', e['of_which_ms']['records_to_float64_rows_on_host_threads'], 'ms/step', r['ms_per_step'])
PY
