#!/bin/bash
# round-3 GPU session C: A/B of the fast vertex placement, instruction mix after pruning, phase profile
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for v in fv0 fv1; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 > $O/models_${v}_$rep.txt 2>&1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events'], 'parity', r['parity_check'])
PY
grep -h passes $O/models_*.txt | sort | cut -c1-110
timeout 600 python tools/ophist.py weave:33 weave:27 gearlike:30 knurling:27 > $O/ophist.txt 2>&1
cat $O/ophist.txt
SDF_MESH_PROF=1 timeout 300 python tools/modeltime.py --on-only example:27 weave:33 > $O/prof.txt 2>&1
grep -h "prof\]" $O/prof.txt | tail -24
