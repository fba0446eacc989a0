#!/bin/bash
# r06f: weave's two-pass sampling kernel with ONE sample per lane (8 spilled vector registers instead of 184): weave 2^27 / 2^33, alternating
set -u
cd "$(dirname "$0")/../.."
TAG=${1:-r06f}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1; do
    SDF_MESH_NS1=$v timeout 600 python tools/modeltime.py --on-only weave:24 weave:27 weave:33 knurling:27 > $O/models_ns1${v}_$rep.txt 2>&1
  done
done
grep -H passes $O/models_*.txt | cut -c1-170
