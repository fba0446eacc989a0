#!/bin/bash
# r06e: phase counters of k_mesh and k_mesh2 on the example at 2^24 and 2^27
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06e}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do
SDF_MESH_PROF=1 SDF_MESH2=$v timeout 300 python tools/modeltime.py --on-only example:24 example:27 > $O/prof_m2$v.txt 2>&1
echo "== SDF_MESH2=$v"; grep -a "k_mesh prof\|passes" $O/prof_m2$v.txt | cut -c1-330
done
