#!/bin/bash
# quick look at a build: identity tests, the bench line (twice), per-model call times, phase counters of the float64 job
#   tools/sessions/gpu_quick.sh <tag> [env assignments for the runs, e.g. SDF_DEFER=0]
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-quick}; shift || true
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or texture or one_pass or tail" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; tail -3 $O/t_identity.txt | head -1
for rep in 1 2; do
  env "$@" timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_$rep.txt 2>&1
done
env "$@" timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models.txt 2>&1
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events'], 'parity', r['parity_check'])
PY
grep -h passes $O/models.txt | cut -c1-110
( env "$@" SDF_MESH_PROF=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --sync ) > $O/prof_bench.txt 2> $O/prof_bench.err
grep -a "prof\]" $O/prof_bench.err | sed -n 8,12p | cut -c1-330
