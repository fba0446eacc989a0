#!/bin/bash
# r06g: (1) issue cost of the vector instructions k_mesh is made of (tools/ubench/valu_rates.hip); (2) A/B: the next batch's tape through the
# scalar cache during the emission (lib_warm) against the base
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06g}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/valu_rates valu_rates.hip && timeout 300 /tmp/valu_rates ) > $O/valu_rates.txt 2>&1
cat $O/valu_rates.txt | head -130
bash tools/gpu_abn.sh ${TAG}_ab base warm
for v in base warm; do SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 blobby:30 gearlike:30 weave:27 > $O/models_$v.txt 2>&1; done
grep -H passes $O/models_*.txt | cut -c1-150
