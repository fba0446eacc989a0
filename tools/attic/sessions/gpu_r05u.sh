#!/bin/bash
# r05u: `_estimate_bounds` as 64 single-wave workgroups (k_estimate_bounds_w) against the four-workgroup form of r05h
# (SDF_BOUNDS_WAVES=0), alternating: every bounds test under both, per-model times, the drop-in caller's generate_e2e
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05u; mkdir -p $O; export TMPDIR=/tmp
for w in 1 0; do
  ( time SDF_BOUNDS_WAVES=$w timeout 400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "bounds" ) > $O/tests_w$w.txt 2>&1
  echo "bounds tests waves=$w rc=$?"; grep -a "passed\|failed\|error" $O/tests_w$w.txt | tail -2
done
for rep in 1 2; do
  for w in 1 0; do
    SDF_BOUNDS_WAVES=$w timeout 200 python tools/boundstime.py > $O/time_w${w}_$rep.txt 2>&1
    echo "waves=$w rep=$rep: $(awk '{printf "%s %s | ", $1, $2}' $O/time_w${w}_$rep.txt)"
  done
done
for w in 1 0; do
  SDF_BOUNDS_WAVES=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_w$w.txt 2>&1
  python - "$O/bench_w$w.txt" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        r=json.loads(l); print(sys.argv[1].split('/')[-1], 'ms/step', r['ms_per_step'], 'e2e', r.get('generate_e2e'), 'parity', r['parity_check'])
PY
done
