#!/bin/bash
# r04a (prepared at the end of round 3, never run): the third culling level on a GPU for the first time.
#   1. the on / off identity tests and the ragged / degenerate grids (bit-exact soups with the interval passes on and off)
#   2. the whole GPU suite
#   3. the bench line and the per-model call times, to compare with profiles/r03x_bench.json (0.3103 ms / step, k_mesh 0.2746 ms,
#      n_sampled / n_eval 22.7 % at C2) -- tools/cull3study.py predicts ~10 % sampled
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or texture" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; tail -3 $O/t_identity.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
( timeout 400 python bench.py ) > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt | cut -c1-700
( timeout 300 python tools/cullstat.py ) 2>&1 | grep -v amdgpu | tail -8 | tee $O/cullstat.txt
