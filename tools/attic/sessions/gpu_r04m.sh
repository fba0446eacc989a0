#!/bin/bash
# r04m: rocprofv3 kernel statistics + PMC passes of this build: the headline command (four calls in flight), one call at a
# time, and weave 2^33
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04m
mkdir -p $O
TAG=${1:-r04m}
bash tools/profile.sh ${TAG} --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/prof_pipe.log 2>&1
bash tools/profile.sh ${TAG}_sync --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --sync > $O/prof_sync.log 2>&1
bash tools/profile.sh ${TAG}_weave33 --model weave --samples-log2 33 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --sync > $O/prof_weave.log 2>&1
find gpurun_out/prof_${TAG}* -name '*kernel_trace.csv' -size +8M -delete
find gpurun_out/prof_${TAG}* -name '*counter_collection.csv' -size +8M -delete
du -sh gpurun_out/prof_${TAG}*
tail -3 $O/prof_sync.log
