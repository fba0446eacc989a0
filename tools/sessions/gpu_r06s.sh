#!/bin/bash
# r06s: gearlike inside the default bench line: allocation trace (SDF_POOL_TRACE) + progress markers around its measurement
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06s}
mkdir -p $O
export TMPDIR=/tmp
SDF_POOL_TRACE=1 SDF_BENCH_TRACE=1 SDF_BENCH_WHOLE_SOUP_S=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/default.txt 2> $O/default.err
grep -n "other config\|measure \|sdf pool" $O/default.err | awk '/other config gearlike/{p=1} /other config blobby/{p=0} p' | cut -c1-160 | awk '{c[$0]++} END{}1' | uniq -c | head -80
