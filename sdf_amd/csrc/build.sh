#!/bin/sh
# Builds libsdf_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -ffp-contract=off: the interpreter must round like NumPy (separate multiply and add);
# fused multiply-adds are written explicitly where the reference goes through BLAS.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
exec "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wno-unused-result -o libsdf_hip.so sdf_hip.hip "$@"
