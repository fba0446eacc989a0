#!/bin/bash
# A/B of two builds of the library (ablibs/lib_old.so, ablibs/lib_new.so): bench + model times, alternating
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-ab}
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for v in old new; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models_${v}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events'], 'parity', r['parity_check'])
PY
grep -h passes $O/models_*.txt | sort | cut -c1-100
