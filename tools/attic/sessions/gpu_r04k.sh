#!/bin/bash
# r04k: whole GPU suite on the build with deferred emission / early work items / packed sub-group states, then the bench and
# the models with k_cull_lean at seven (default) and six (SDF_CULL_LDS_CAP=0) workgroups per CU
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04k
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
bash tools/sessions/gpu_quick.sh r04k_cap1 | grep "bench_\|passes"
bash tools/sessions/gpu_quick.sh r04k_cap0 SDF_CULL_LDS_CAP=0 | grep "bench_\|passes"
