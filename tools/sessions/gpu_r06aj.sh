#!/bin/bash
# r06aj: three samples per lane for the remaining register files: pawn (2,4), weave (4,2; two-pass), and forced one- / two-pass for the others
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06aj}
mkdir -p $O
export TMPDIR=/tmp
SDF_MESH_SHAPE=3 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "generate_matches or one_pass_and_two_pass or weave or pawn" 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2 3; do for sh in 0 3; do SDF_MESH_SHAPE=$sh timeout 300 python tools/modeltime.py --on-only pawn:27 weave:24 weave:27 weave:33 > $O/models_shape${sh}_$rep.txt 2>&1; done; done
grep -H passes $O/models_*.txt | cut -c1-150 | sort -k2,3 -s | awk '{print $1, $2, $3, $5, $6, $7, $8, $9, $10, $11, $12}' | cut -c22-140
