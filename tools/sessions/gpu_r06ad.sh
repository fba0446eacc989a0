#!/bin/bash
# r06ad: weave's two-pass sampling kernel as 512 threads x FOUR samples per lane (256 VGPRs, 29 spilled instead of 184; half the scalar stream per sample):
# SDF_MESH_SHAPE=1 against the default 1024 x 2, one box, alternating
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06ad}
mkdir -p $O
export TMPDIR=/tmp
SDF_MESH_SHAPE=1 timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "weave" 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do for sh in 0 1; do SDF_MESH_SHAPE=$sh timeout 300 python tools/modeltime.py --on-only weave:24 weave:27 weave:33 > $O/models_shape${sh}_$rep.txt 2>&1; done; done
grep -H passes $O/models_*.txt | cut -c1-150
