#!/bin/bash
# r04w: the parity suite under the NON-default schemes: three interval levels for every tape (trig / rare-leaf tapes take two
# by default), dense tiles + parking for every batch, and the full-size tests with three levels
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04w
mkdir -p $O
export TMPDIR=/tmp
( time SDF_CULL_LEVELS=3 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q ) > $O/t_levels3.txt 2>&1
echo "levels=3 rc=$?"; grep -a "passed\|failed" $O/t_levels3.txt | tail -1
( time SDF_DEFER=0 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q ) > $O/t_defer0.txt 2>&1
echo "defer=0 rc=$?"; grep -a "passed\|failed" $O/t_defer0.txt | tail -1
( time SDF_CULL_LEVELS=2 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or texture or one_pass or tail or fixture" ) > $O/t_levels2.txt 2>&1
echo "levels=2 rc=$?"; grep -a "passed\|failed" $O/t_levels2.txt | tail -1
( time SDF_CULL_LEVELS=3 timeout 1200 python -m pytest tests/test_full_size.py -m gpu -x -q ) > $O/t_full_levels3.txt 2>&1
echo "full-size levels=3 rc=$?"; grep -a "passed\|failed" $O/t_full_levels3.txt | tail -1
