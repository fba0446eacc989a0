#!/bin/bash
# round 5: what does a COLD instruction cache cost k_mesh and k_cull?  Each launched twice in a row (SDF_MESH_TWICE / SDF_CULL_TWICE),
# synchronous calls, rocprofv3 kernel trace: durations of the first and the second launch of every pair.
set -u
cd "$(dirname "$0")/../.."
REPO=$PWD; O=$REPO/gpurun_out/r05k; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
SDF_MESH_TWICE=1 SDF_CULL_TWICE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $REPO/bench.py --steps 30 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-check > $O/trace.log 2>&1
echo "rc=$?"
python - "$O" <<'PY'
import csv,glob,sys,statistics as st
rows=[]
for f in glob.glob(sys.argv[1]+'/trace/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for name in ('k_mesh','k_cull'):
    ks=[r for r in rows if name in r['Kernel_Name']]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in ks]
    first, second = d[0::2], d[1::2]
    n=min(len(first),len(second))
    print(name, 'pairs', n, 'first (cold) median %.1f us  second (warm) median %.1f us' % (st.median(first[5:n]), st.median(second[5:n])), ' min %.1f / %.1f' % (min(first[5:n]), min(second[5:n])))
PY
