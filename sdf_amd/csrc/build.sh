#!/bin/sh
# Builds libsdf_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -ffp-contract=off: the interpreter must round like NumPy (separate multiply and add);
# fused multiply-adds are written explicitly where the reference goes through BLAS.
# The fused sample+march kernel is instantiated per (precision, trig) family in its own
# translation unit so the families compile in parallel.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -structurizecfg-skip-uniform-regions: every branch of the interpreter is wave-uniform; without it
# the backend structurises the dispatch tree anyway and pays ~100 register copies per tape instruction
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -structurizecfg-skip-uniform-regions=1"
mkdir -p build
rm -f libsdf_hip.so build/*.o
pids=""
$HIPCC $FLAGS -c -o build/sdf_hip.o sdf_hip.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_T=double -DMESH_FULL=0 -DMESH_NAME=sdf_launch_mesh_f64 -c -o build/mesh_f64.o sdf_mesh_inst.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_T=double -DMESH_FULL=1 -DMESH_NAME=sdf_launch_mesh_f64_full -c -o build/mesh_f64_full.o sdf_mesh_inst.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_T=float -DMESH_FULL=0 -DMESH_NAME=sdf_launch_mesh_f32 -c -o build/mesh_f32.o sdf_mesh_inst.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_T=float -DMESH_FULL=1 -DMESH_NAME=sdf_launch_mesh_f32_full -c -o build/mesh_f32_full.o sdf_mesh_inst.hip "$@" & pids="$pids $!"
# (the weld uses hipCUB's radix sort and scan; it has no floating-point arithmetic of its own)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c -o build/sdf_weld.o sdf_weld.hip "$@" & pids="$pids $!"
for p in $pids; do wait $p; done
exec $HIPCC --offload-arch=gfx950 -fPIC -shared -o libsdf_hip.so build/sdf_hip.o build/mesh_f64.o build/mesh_f64_full.o \
    build/mesh_f32.o build/mesh_f32_full.o build/sdf_weld.o
