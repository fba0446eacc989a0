#!/bin/bash
# r05v: where the prepass goes for the other BASELINE models (rocprofv3 kernel statistics of one synchronous call each, 3 repeats)
set -u
cd "$(dirname "$0")/../.."
REPO=$(pwd); O=$REPO/gpurun_out/r05v; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for job in blobby:30 gearlike:30 pawn:27 knurling:27; do
  n=${job%%:*}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o $n -- python $REPO/tools/modeltime.py --on-only $job > $O/$n.log 2>&1
  grep passes $O/$n.log | cut -c1-200
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -12 "$f" | cut -d, -f1-7 | cut -c1-200
  find $O/$n -name '*kernel_trace.csv' -delete
done
