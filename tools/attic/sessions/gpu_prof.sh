#!/bin/bash
# phase counters only (SDF_MESH_PROF=1): tools/sessions/gpu_prof.sh <tag> [env...]
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-prof}; shift || true
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( env "$@" SDF_MESH_PROF=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --sync ) > $O/prof_bench.txt 2> $O/prof_bench.err
grep -a "prof\]" $O/prof_bench.err | sed -n 8,14p | cut -c1-330
env "$@" timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_1.txt 2>&1
grep -h '^{"metric"' $O/bench_1.txt | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events'], 'parity', r['parity_check'])"
