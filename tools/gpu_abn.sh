#!/bin/bash
# A/B/... of several builds of the library (ablibs/lib_<name>.so for every name given): bench line (k_mesh alone, step), alternating
#   tools/gpu_abn.sh <tag> name1 name2 ...
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in "$@"; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_${v}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'parity', r['parity_check'])
PY
