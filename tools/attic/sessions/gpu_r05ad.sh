#!/bin/bash
# r05ad: the look-back through block sums (one round trip: count words of the own block + block words of the own group + group words)
# against the walk over status words (ablibs/lib_base.so = r05ab): identity / scheme / shard / exchange tests on the new library, then
# bench (200 steps, sustained, isolated k_mesh) and per-model call times, alternating
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r05ad}; mkdir -p $O; export TMPDIR=/tmp
( time SDF_HIP_LIB=$PWD/ablibs/lib_lb.so timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "golden or shard or lattice or park" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; grep -a "passed\|failed\|error" $O/t_identity.txt | tail -2
for rep in 1 2 3; do
  for v in base lb; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
  done
done
for rep in 1; do for v in base lb; do
  SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 > $O/models_${v}_$rep.txt 2>&1
done; done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], r['isolated_calls']['k_mesh_ms_device_clock']['median'], 'parity', r['parity_check'])
PY
grep -H passes $O/models_*.txt | sed 's/.*models_//' | cut -c1-112 | sort -k2,3 -s
