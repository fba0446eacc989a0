#!/bin/bash
# r04a: the third culling level (branch next-cull3, merged) on a GPU for the first time.
#   1. the on / off identity tests and the ragged / degenerate grids (bit-exact soups with the interval passes on and off)
#   2. the whole GPU suite
#   3. A/B against the round-3 library (ablibs/lib_old.so): bench line + per-model call times, alternating
#   4. phase counters of the new build (SDF_MESH_PROF=1) and the sampled share per model
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or texture" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; tail -3 $O/t_identity.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
bash tools/gpu_ab.sh r04a_ab 2>&1 | tail -40
( SDF_MESH_PROF=1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --sync ) > $O/prof_bench.txt 2> $O/prof_bench.err
grep -a "prof\]" $O/prof_bench.err | tail -12
( timeout 300 python tools/cullstat.py ) 2>&1 | grep -v amdgpu | tail -8 | tee $O/cullstat.txt
( timeout 400 python bench.py ) > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt | cut -c1-900
