#!/bin/bash
# r06w: what does the library do during the one slow submission inside gearlike's timed steps?  (pool trace between the progress markers)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06w}
mkdir -p $O
export TMPDIR=/tmp
SDF_BENCH_OTHER_ORDER=gearlike SDF_BENCH_WHOLE_SOUP_S=1 SDF_BENCH_TRACE=1 SDF_POOL_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check > $O/run.txt 2> $O/run.err
awk '/measure gearlike 2\^30: 24 steps, 4/{p=1} /measure gearlike done/{if(p)print; p=0} p' $O/run.err | grep -v "collected" | cut -c1-110 | tail -75
