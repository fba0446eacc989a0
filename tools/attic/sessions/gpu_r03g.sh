#!/bin/bash
# round-3 GPU session G: whole GPU suite, the bench line, rocprofv3 profiles of the headline config (pipelined and one
# call at a time) and of weave 2^33
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -5 $O/tests.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench.err
echo "bench rc=$?"
bash tools/profile.sh r03g --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/prof_pipe.log 2>&1
bash tools/profile.sh r03g_sync --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --sync > $O/prof_sync.log 2>&1
bash tools/profile.sh r03g_weave33 --model weave --samples-log2 33 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --sync > $O/prof_weave.log 2>&1
( time timeout 300 python tools/disttime.py 30 ) > $O/disttime.txt 2>&1
grep chunks $O/disttime.txt
# keep the merge small: drop the per-dispatch traces, keep stats and counters
find gpurun_out/prof_r03g* -name '*kernel_trace.csv' -size +8M -delete
du -sh gpurun_out/prof_r03g*
