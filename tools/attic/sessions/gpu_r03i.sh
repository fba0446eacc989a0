#!/bin/bash
# r03i: the build with every non-interpreter kernel in sdf_plain.hip (no structurizer option) and the streaming k_expand:
# the whole GPU suite, the multi-GPU step's stage times, the default bench line.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $O/tests.txt
( timeout 200 python tools/disttime.py ) > $O/disttime.txt 2>&1; grep -v amdgpu $O/disttime.txt | tail -12 | cut -c1-200
( timeout 400 python bench.py ) > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt | cut -c1-1500
