#!/bin/bash
# r06y: k_cull's phase counters (SDF_MESH_PROF=1) at 512^3: where does the "start" of a workgroup go?
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06y}
mkdir -p $O
export TMPDIR=/tmp
SDF_MESH_PROF=1 timeout 300 python tools/modeltime.py --on-only example:27 > $O/prof.txt 2>&1
grep "prof\]" $O/prof.txt | tail -9
