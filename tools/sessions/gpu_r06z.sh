#!/bin/bash
# r06z: float64 selects through sel64 (VOP3 v_cndmask with the condition in a scalar pair, sdf_vec.h) + the square root written out:
# identity tests with the new library, then A/B against the same source with -DSDF_SEL64=0 (bench line + models, alternating)
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06z}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
SDF_HIP_LIB=$PWD/ablibs/lib_sel.so timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "values_match or generate_matches or interval_passes_identity or marching_cubes or bounds" 2>&1 | tail -4 | tee $O/tests.txt
bash tools/gpu_abn.sh ${TAG}_ab base sel | tee $O/ab.txt
for rep in 1 2; do for v in base sel; do SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 blobby:30 gearlike:30 knurling:27 weave:27 weave:33 > $O/models_${v}_$rep.txt 2>&1; done; done
grep -H passes $O/models_*.txt | cut -c1-140
