#!/bin/bash
# round 5: k_mesh with its argument block read from the kernel-argument segment on demand (SGPR spills 201 -> 131) against the
# committed build, alternating; k_cull with two waves per work item (SDF_CULL_BLOCK=128) against the default.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05i; mkdir -p $O; export TMPDIR=/tmp
( SDF_HIP_LIB=$PWD/ablibs/lib_kernarg.so timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "schemes or deferred or one_pass or interval or golden or generate_matches" ) > $O/t_kernarg.txt 2>&1
echo "tests (kernarg) rc=$?"; tail -1 $O/t_kernarg.txt
for rep in 1 2 3; do
  for v in main kernarg; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
  done
done
for rep in 1 2; do
  for cb in 0 128; do
    SDF_HIP_LIB=$PWD/ablibs/lib_main.so SDF_CULL_BLOCK=$cb timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_cull${cb}_$rep.txt 2>&1
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
for v in main kernarg; do
  SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models_$v.txt 2>&1
  echo "== $v"; grep -h passes $O/models_$v.txt | cut -c1-100
done
