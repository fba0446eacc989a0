#!/bin/bash
# r06b: first run of k_mesh2 (two workgroups of 512 threads per CU): its parity test, then bench + model times with SDF_MESH2=0 / 1 alternating
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06b}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "two_workgroups or core_module_seams or batch_size_above_32" ) > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -15 $O/tests.txt
for rep in 1 2; do
  for v in 0 1; do
    SDF_MESH2_DEBUG=1 SDF_MESH2=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_m2${v}_$rep.txt 2> $O/bench_m2${v}_$rep.err
    SDF_MESH2_DEBUG=1 SDF_MESH2=$v timeout 300 python tools/modeltime.py --on-only example:24 example:27 pawn:27 blobby:30 > $O/models_m2${v}_$rep.txt 2>&1
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
grep -H "passes\|k_mesh2" $O/models_*.txt | cut -c1-160
tail -3 $O/bench_m21_1.err
