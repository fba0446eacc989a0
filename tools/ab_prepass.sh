#!/bin/bash
# A/B timing of two builds of the library on the GPU box (gpurun):  tools/ab_prepass.sh
#   sdf_amd/csrc/libsdf_hip_base.so = the build to compare against (copied there by hand), libsdf_hip.so = this tree
# Writes gpurun_out/ab_*.txt: per-model prepass / mesh times (tools/modeltime.py) for every variant, then bench lines.
mkdir -p gpurun_out
JOBS="example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27"
BASE=$PWD/sdf_amd/csrc/libsdf_hip_base.so
run() { echo "== $1"; shift; env "$@" python tools/modeltime.py --on-only $JOBS 2>&1 | grep -v Warning; }
{
  [ -f "$BASE" ] && run "base" SDF_HIP_LIB=$BASE
  run "new cull block 256"
  run "new cull block 64" SDF_CULL_BLOCK=64
  run "new cull block 128" SDF_CULL_BLOCK=128
  run "new cull block 512" SDF_CULL_BLOCK=512
  [ -f "$BASE" ] && { echo "== base weave 2^33"; SDF_HIP_LIB=$BASE python tools/modeltime.py --on-only weave:33 2>&1 | grep -v Warning; }
  echo "== new weave 2^33"; python tools/modeltime.py --on-only weave:33 2>&1 | grep -v Warning
} > gpurun_out/ab_models.txt 2>&1
{
  [ -f "$BASE" ] && { echo "== base"; SDF_HIP_LIB=$BASE python bench.py --no-cpu-baseline 2>&1 | tail -1; }
  echo "== new"; python bench.py --no-cpu-baseline 2>&1 | tail -1
  echo "== new, cull block 128"; SDF_CULL_BLOCK=128 python bench.py --no-cpu-baseline 2>&1 | tail -1
  echo "== new, cull block 64"; SDF_CULL_BLOCK=64 python bench.py --no-cpu-baseline 2>&1 | tail -1
} > gpurun_out/ab_bench.txt 2>&1
