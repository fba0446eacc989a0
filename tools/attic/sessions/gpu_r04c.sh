#!/bin/bash
# r04c: deferred emission with the wide look-back, the transposed (coalesced) soup stores and room in the park FIFO for
# what a round may park; phase counters of the float64 job; bench + models; then the GPU suite
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04c
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or texture or one_pass or tail" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; tail -3 $O/t_identity.txt
( time timeout 900 python -m pytest tests/test_full_size.py -m gpu -x -q ) > $O/t_full.txt 2>&1
echo "full-size rc=$?"; tail -3 $O/t_full.txt
for rep in 1 2; do
for cfg in "1 0" "0 0"; do
  set -- $cfg
  SDF_DEFER=$1 SDF_CULL_LEVELS=$2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_d$1_l$2_$rep.txt 2>&1
  SDF_DEFER=$1 SDF_CULL_LEVELS=$2 timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models_d$1_l$2_$rep.txt 2>&1
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
grep -H passes $O/models_*.txt | sed 's/.*models_//' | sort -k2,3 | cut -c1-125
( SDF_MESH_PROF=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --sync ) > $O/prof_bench.txt 2> $O/prof_bench.err
grep -a "prof\]" $O/prof_bench.err | sed -n 7,12p | cut -c1-330
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_full_size.py ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
