#!/bin/bash
# round 5: k_mesh whose waves work out a first chunk of the waiting batch's vertices while wave 0 walks the look-back (lib_overlap)
# against the committed build (lib_main), alternating; identity tests under the new library first.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05m; mkdir -p $O; export TMPDIR=/tmp
( SDF_HIP_LIB=$PWD/ablibs/lib_overlap.so timeout 400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "schemes or deferred or one_pass or interval or golden or generate_matches or rare_paths or tail or park or slab or exchange_emulated" ) > $O/t_overlap.txt 2>&1
echo "tests (overlap) rc=$?"; tail -1 $O/t_overlap.txt
( SDF_HIP_LIB=$PWD/ablibs/lib_overlap.so timeout 300 python -m pytest tests/test_full_size.py -m gpu -x -q -k "matches_oracle_and_reference" ) > $O/t_full_overlap.txt 2>&1
echo "full size (overlap) rc=$?"; tail -1 $O/t_full_overlap.txt
for rep in 1 2 3; do
  for v in main overlap; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events'], 'parity', r['parity_check'])
PY
for v in main overlap; do
  SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only pawn:27 knurling:27 blobby:30 gearlike:30 > $O/models_$v.txt 2>&1
  echo "== $v"; grep -h passes $O/models_$v.txt | cut -c1-100
done
