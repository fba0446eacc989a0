#!/bin/bash
# r05ab: three small changes to k_mesh, one library each (ablibs/lib_{base,m1,m2,m4,m124}.so): M1 = the block scans of the count phase with ONE
# barrier (two alternating buffers), M2 = the look-back's wave sum through the vector ALU's row shifts instead of twelve ds_bpermute,
# M4 = the sign bits cleared when a batch has been counted instead of in front of a barrier of its own at the next batch's start
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r05ab}; mkdir -p $O; export TMPDIR=/tmp
for v in m124 m1; do
( time SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or one_pass or tail or golden or reference or shard or two_pass or batch_size or lattice" ) > $O/t_identity_$v.txt 2>&1
echo "identity $v rc=$?"; grep -a "passed\|failed\|error" $O/t_identity_$v.txt | tail -2
done
for rep in 1 2 3; do
  for v in base m1 m2 m4 m124; do
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_${v}_$rep.txt 2>&1
  done
done
for v in base m124; do
  SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 > $O/models_${v}.txt 2>&1
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], r['isolated_calls']['k_mesh_ms_device_clock']['median'], 'parity', r['parity_check'])
PY
grep -H passes $O/models_*.txt | sed 's/.*models_//' | cut -c1-112 | sort -k2,3 -s
