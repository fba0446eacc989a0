#!/bin/bash
# r06a: baseline of the round-5 source on this round's boxes: whole GPU suite, the driver's bench line, k_cull's histogram of listed tasks per work item
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06a}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.txt | cut -c1-600
SDF_MESH_PROF=1 timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 > $O/prof_models.txt 2>&1
grep -a "listed tasks\|passes" $O/prof_models.txt | cut -c1-200
