#!/bin/sh
# build + run every micro-benchmark in this directory on the GPU box
cd "$(dirname "$0")"
for f in *.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/${f%.hip} $f && /tmp/${f%.hip}
done
