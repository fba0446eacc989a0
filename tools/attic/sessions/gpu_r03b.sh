#!/bin/bash
# round-3 GPU session B: whole GPU suite on the new build (spin waits, device clocks, comm, sincos64), model times
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python tools/modeltime.py --on-only example:27 gearlike:30 blobby:30 weave:27 weave:33 knurling:27 pawn:27 ) > $O/models.txt 2>&1
echo "models rc=$?"; cat $O/models.txt | grep passes
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -8 $O/tests.txt
