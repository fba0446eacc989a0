#!/bin/bash
# round 5: k_cull computing its tile's axis values (np.arange's fill: first + i * delta, checked on the host) instead of loading them
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05l; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or cull or ragged or edge or generate_matches or schemes or deferred" ) > $O/t.txt 2>&1
echo "tests rc=$?"; tail -1 $O/t.txt
for rep in 1 2 3; do
  for ar in 1 0; do
    SDF_AXES_ARITH=$ar timeout 200 python bench.py --steps 100 --warmup 10 --sync --no-cpu-baseline --no-other-configs > $O/bench_sync_arith${ar}_$rep.txt 2>&1
  done
done
for ar in 1 0; do SDF_AXES_ARITH=$ar timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_pipe_arith${ar}.txt 2>&1; done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'prepass', r['isolated_calls']['prepass_ms'], 'parity', r['parity_check'])
PY
for ar in 1 0; do
  SDF_AXES_ARITH=$ar timeout 300 python tools/modeltime.py --on-only pawn:27 blobby:30 gearlike:30 weave:33 > $O/models_arith$ar.txt 2>&1
  echo "== SDF_AXES_ARITH=$ar"; grep -h passes $O/models_arith$ar.txt | cut -c1-100
done
