#!/bin/bash
# r05z: (1) blobby 2^30 with four calls in flight: 0.886 ms per step in r05s, 1.519 in r05y -- which build, or the box?  three builds,
# alternating; (2) k_compact / k_scan_items with eight elements per thread and pass (lib_new) against lib_old: per-model call times
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r05z}; mkdir -p $O; export TMPDIR=/tmp
( time SDF_HIP_LIB=$PWD/ablibs/lib_new.so timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or one_pass or tail or golden or reference or shard or two_pass or batch_size" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; grep -a "passed\|failed\|error" $O/t_identity.txt | tail -2
for rep in 1 2; do
  for v in r05s old new; do
    for d in 4 1; do
      SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python bench.py --model blobby --samples-log2 30 --steps 24 --warmup 2 --inflight $d --no-cpu-baseline --no-other-configs > $O/blobby_${v}_d${d}_$rep.txt 2>&1
    done
    SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 blobby:30 gearlike:30 weave:33 > $O/models_${v}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/blobby_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r.get('latency_ms_per_call'), 'parity', r.get('parity_check'))
PY
grep -H passes $O/models_*.txt | sed 's/.*models_//' | cut -c1-112 | sort -k2,3 -s
