#!/bin/bash
# r06ae: kernel statistics + counters (pipelined, one call at a time, weave 2^33) and the default bench line of the source handed in (the drop-in sections are left out of the profile passes: their k_mesh launches write 16-byte records)
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06ae}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
bash tools/profile.sh ${TAG} --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/prof_pipe.log 2>&1
bash tools/profile.sh ${TAG}_sync --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --sync > $O/prof_sync.log 2>&1
bash tools/profile.sh ${TAG}_weave33 --model weave --samples-log2 33 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --sync > $O/prof_weave.log 2>&1
find gpurun_out/prof_${TAG}* -name '*kernel_trace.csv' -size +8M -delete
find gpurun_out/prof_${TAG}* -name '*counter_collection.csv' -size +8M -delete
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.txt | cut -c1-300
