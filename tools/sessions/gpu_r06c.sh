#!/bin/bash
# r06c: k_mesh2 debugging: which limit do flagged tiles meet; its parity test
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06c}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
SDF_MESH2_DEBUG=1 SDF_MESH2=1 timeout 300 python tools/modeltime.py --on-only example:24 example:27 pawn:27 blobby:30 gearlike:30 knurling:27 > $O/models_m21.txt 2>&1
grep -a "passes\|k_mesh2" $O/models_m21.txt | cut -c1-150
( time SDF_MESH2_DEBUG=1 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "two_workgroups or core_module_seams" ) > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -25 $O/tests.txt | cut -c1-200
