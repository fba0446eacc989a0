#!/bin/bash
# r05w: k_cull for the lean tapes with eight waves per work item and / or without the level of 8^3-cell boxes (all 512 groups
# of 4^3 cells evaluated at once): identity tests under the new form, then bench + per-model times, alternating
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05w; mkdir -p $O; export TMPDIR=/tmp
( time SDF_CULL_BLOCK=512 SDF_CULL_SKIP0=1 timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or one_pass or tail or golden or reference" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; grep -a "passed\|failed\|error" $O/t_identity.txt | tail -2
for rep in 1 2; do
  for v in "256 0" "512 1" "512 0" "256 1"; do
    set -- $v
    SDF_CULL_BLOCK=$1 SDF_CULL_SKIP0=$2 timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_b$1_s$2_$rep.txt 2>&1
    SDF_CULL_BLOCK=$1 SDF_CULL_SKIP0=$2 timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 blobby:30 > $O/models_b$1_s$2_$rep.txt 2>&1
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
grep -H passes $O/models_*.txt | sed 's/.*models_//' | cut -c1-150
