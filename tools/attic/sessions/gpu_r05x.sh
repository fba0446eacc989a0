#!/bin/bash
# r05x: A/B of two builds (ablibs/lib_old.so, ablibs/lib_new.so): identity tests on the new one, then bench (200 steps: sustained,
# isolated prepass / k_mesh) and per-model times, alternating
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r05x}; mkdir -p $O; export TMPDIR=/tmp
( time SDF_HIP_LIB=$PWD/ablibs/lib_new.so timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or one_pass or tail or golden or reference" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; grep -a "passed\|failed\|error" $O/t_identity.txt | tail -2
for rep in 1 2; do
  for v in old new; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models_${v}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'prepass', r['isolated_calls']['prepass_ms']['median'], 'parity', r['parity_check'])
PY
grep -H passes $O/models_*.txt | sed 's/.*models_//' | cut -c1-112 | sort -k2,3 -s
