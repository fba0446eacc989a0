#!/bin/bash
# r06i: A/B: k_mesh (the family without trigonometry) with every v_cndmask_b32 in its VOP3 encoding (lib_e64) against the same source as compiled (lib_warm)
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06i}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
bash tools/gpu_abn.sh ${TAG}_ab warm e64
for rep in 1 2; do for v in warm e64; do SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 blobby:30 > $O/models_${v}_$rep.txt 2>&1; done; done
grep -H passes $O/models_*.txt | cut -c1-150
SDF_HIP_LIB=$PWD/ablibs/lib_e64.so timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "generate_matches or sparse_tiles or two_workgroups" 2>&1 | tail -3
