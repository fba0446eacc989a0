#!/bin/bash
# r06ak: the source handed in (round 6, three samples per lane in the one-pass kernels): whole GPU suite, the bench line as the driver runs it, rocprofv3 kernel statistics +
# PMC passes (headline pipelined, one call at a time, weave 2^33), the one-rank exchange step
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06ak}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.txt | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile.sh ${TAG} --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/prof_pipe.log 2>&1
bash tools/profile.sh ${TAG}_sync --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --sync > $O/prof_sync.log 2>&1
bash tools/profile.sh ${TAG}_weave33 --model weave --samples-log2 33 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --sync > $O/prof_weave.log 2>&1
find gpurun_out/prof_${TAG}* -name '*kernel_trace.csv' -size +8M -delete
find gpurun_out/prof_${TAG}* -name '*counter_collection.csv' -size +8M -delete
( timeout 300 python tools/disttime.py 30 ) > $O/disttime.txt 2>&1
grep chunks $O/disttime.txt
