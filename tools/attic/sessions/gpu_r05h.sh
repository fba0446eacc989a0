#!/bin/bash
# round 5: k_estimate_bounds variants (SDF_BOUNDS_MODE 1 = one workgroup NS 4; 2 = four workgroups, a probe per lane, one-word exchange;
# 3 = two workgroups NS 2) per model, and the 116 reference bounds under each
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp
for mode in 2 3 1; do
  echo "== SDF_BOUNDS_MODE=$mode"
  SDF_BOUNDS_MODE=$mode timeout 120 python tools/boundstime.py 2>&1 | tail -6
  SDF_BOUNDS_MODE=$mode timeout 200 python -m pytest tests/test_gpu.py -m gpu -x -q -k "bounds" 2>&1 | tail -1
done
