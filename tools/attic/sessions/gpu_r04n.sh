#!/bin/bash
# r04n: the 16-byte slab records: every exchange test (synthetic slabs, emulated ranks, processes through the stand-in
# librccl, real RCCL with one rank), the full-size tests (8 / 16 emulated slabs at C2 .. C4), then the rest of the suite,
# the bench line and the stage times of the one-rank exchange step
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04n
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "slab or exchange or expand or sharded or bench_two or native or skip_test" ) > $O/t_exchange.txt 2>&1
echo "exchange rc=$?"; tail -3 $O/t_exchange.txt | head -1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
bash tools/sessions/gpu_quick.sh r04n_q | grep "bench_\|passes"
( timeout 300 python tools/disttime.py 30 ) > $O/disttime.txt 2>&1
grep chunks $O/disttime.txt
