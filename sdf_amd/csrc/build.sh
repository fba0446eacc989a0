#!/bin/sh
# Builds libsdf_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -ffp-contract=off: the interpreter must round like NumPy (separate multiply and add);
# fused multiply-adds are written explicitly where the reference goes through BLAS.
# The fused sample+march kernel is instantiated per family (float64, with / without the trigonometric ops) in its own
# translation unit so the families compile in parallel.  (float32 sampling of the meshing path: removed in round 5.)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -structurizecfg-skip-uniform-regions: every branch of the tape interpreter is wave-uniform and its dispatch is a
# scalar jump through a table (asm goto + s_setpc, sdf_interp.h); the structurizer has to leave that alone -- a build of
# the interpreters WITHOUT the option faults on the device.  The option is NOT safe for kernels whose lanes diverge
# (r03: it miscompiled k_expand), so only the interpreters' translation units get it and every other kernel lives in
# sdf_plain.hip / sdf_weld.hip.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -structurizecfg-skip-uniform-regions=1"
mkdir -p build
rm -f libsdf_hip.so build/*.o
# what built the library goes INTO the library (sdf_build_info(): compiler version + the interpreters' flags): a box without the
# test suite can still say which toolchain its .so came from; tools/isa_check.py (run by build()) checks what that toolchain made
INFO="$($HIPCC --version | grep -m1 -i 'HIP version' | tr -d '"' | sed 's/^ *//'); $($HIPCC --version | grep -m1 -i 'clang version' | tr -d '"' | cut -c1-60); flags: $FLAGS"
pids=""
$HIPCC $FLAGS "-DSDF_BUILD_INFO=\"$INFO\"" -c -o build/sdf_hip.o sdf_hip.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -c -o build/sdf_bounds.o sdf_bounds.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_T=double -DMESH_FULL=0 -DMESH_NAME=sdf_launch_mesh_f64 -c -o build/mesh_f64.o sdf_mesh_inst.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_T=double -DMESH_FULL=1 -DMESH_NAME=sdf_launch_mesh_f64_full -c -o build/mesh_f64_full.o sdf_mesh_inst.hip "$@" & pids="$pids $!"
# (k_mesh2, sdf_mesh2.h: the same rounds as two workgroups of 512 threads per compute unit)
$HIPCC $FLAGS -DMESH_FULL=0 -DMESH_NAME=sdf_launch_mesh2_f64 -c -o build/mesh2_f64.o sdf_mesh2_inst.hip "$@" & pids="$pids $!"
$HIPCC $FLAGS -DMESH_FULL=1 -DMESH_NAME=sdf_launch_mesh2_f64_full -c -o build/mesh2_f64_full.o sdf_mesh2_inst.hip "$@" & pids="$pids $!"
# (every kernel that is not a tape interpreter: built WITHOUT the structurizer option, see sdf_plain.hip)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -c -o build/sdf_plain.o sdf_plain.hip "$@" & pids="$pids $!"
# (the weld uses hipCUB's radix sort and scan; it has no floating-point arithmetic of its own)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c -o build/sdf_weld.o sdf_weld.hip "$@" & pids="$pids $!"
for p in $pids; do wait $p; done
exec $HIPCC --offload-arch=gfx950 -fPIC -shared -o libsdf_hip.so build/sdf_hip.o build/mesh_f64.o build/mesh_f64_full.o build/mesh2_f64.o build/mesh2_f64_full.o \
    build/sdf_bounds.o build/sdf_weld.o build/sdf_plain.o
