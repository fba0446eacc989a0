#!/bin/bash
# r03j: rocprofv3 kernel statistics of the final build: the headline config one call at a time, and the multi-GPU step
# with a one-rank communicator (tools/disttime.py: k_pack_slab / k_expand / k_collect_headers next to the meshing kernels).
set -u
cd "$(dirname "$0")/../.."
REPO=$(pwd)
O=$REPO/gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sync -o sync -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --sync > $O/sync.log 2>&1
echo "sync rc=$?"; grep -h '^{"metric"' $O/sync.log | tail -1 | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/exchange -o exchange -- python $REPO/tools/disttime.py 40 > $O/exchange.log 2>&1
echo "exchange rc=$?"; grep "in flight" $O/exchange.log
find $O -name '*kernel_trace.csv' -delete
find $O -name '*.csv' | head
