#!/bin/bash
# r04aa: circular_array's sector by a binary search of rotations (no atan2 / sincos): value fixtures and sector-boundary
# goldens, identity tests, full-size tolerance tests (gearlike 2^30, weave 2^24 / 2^33), per-model times
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04aa
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "values or circ or bounds or interval or prune or cull or ragged or random_csg or arrays or trig or golden or generate" ) > $O/t_values.txt 2>&1
echo "values rc=$?"; grep -a "passed\|failed" $O/t_values.txt | tail -1
( time timeout 1200 python -m pytest tests/test_full_size.py -m gpu -x -q ) > $O/t_full.txt 2>&1
echo "full-size rc=$?"; grep -a "passed\|failed" $O/t_full.txt | tail -1
timeout 300 python tools/modeltime.py --on-only example:27 gearlike:30 weave:27 weave:33 knurling:27 2>&1 | grep passes | cut -c1-125
