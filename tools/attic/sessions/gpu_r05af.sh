#!/bin/bash
# r05af: the default bench line of the source handed in, with progress markers, under a short leash (the run of r05ae did not return)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05af; mkdir -p $O; export TMPDIR=/tmp
( time SDF_BENCH_TRACE=1 SDF_BENCH_OTHER_TIMEOUT_S=70 timeout 118 python bench.py ) > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"
grep -a "bench rank" $O/bench.err | tail -25 | cut -c1-160
tail -c 300 $O/bench.txt
