#!/bin/bash
# r04l: the GPU suite with durations, weave 2^33 one pass (deferred emission) against two passes, the full bench line
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04l
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3; grep -a -A14 "slowest" $O/tests.txt | cut -c1-150
for tp in 1 0; do
  SDF_MESH_TWOPASS=$tp timeout 300 python tools/modeltime.py --on-only weave:33 weave:27 2>&1 | grep passes | sed "s/^/twopass=$tp /" | cut -c1-130
done
( timeout 600 python bench.py ) > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.txt | cut -c1-1500
