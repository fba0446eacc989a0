#!/bin/sh
# development build: recompile ONLY the named translation units of csrc/ (hip bounds f64 f64full m2 m2full plain weld) and relink;
# build.sh (what build() runs) always rebuilds everything.   tools/devbuild.sh hip plain
set -e
cd "$(dirname "$0")/../sdf_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -structurizecfg-skip-uniform-regions=1"
PLAIN="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
pids=""
for u in "$@"; do
  case $u in
    hip) $HIPCC $FLAGS "-DSDF_BUILD_INFO=\"devbuild: $($HIPCC --version | grep -m1 -i 'HIP version' | tr -d '"' | sed 's/^ *//'); flags: $FLAGS\"" -c -o build/sdf_hip.o sdf_hip.hip & pids="$pids $!" ;;
    bounds) $HIPCC $FLAGS -c -o build/sdf_bounds.o sdf_bounds.hip & pids="$pids $!" ;;
    f64) $HIPCC $FLAGS -DMESH_T=double -DMESH_FULL=0 -DMESH_NAME=sdf_launch_mesh_f64 -c -o build/mesh_f64.o sdf_mesh_inst.hip & pids="$pids $!" ;;
    f64full) $HIPCC $FLAGS -DMESH_T=double -DMESH_FULL=1 -DMESH_NAME=sdf_launch_mesh_f64_full -c -o build/mesh_f64_full.o sdf_mesh_inst.hip & pids="$pids $!" ;;
    m2) $HIPCC $FLAGS -DMESH_FULL=0 -DMESH_NAME=sdf_launch_mesh2_f64 -c -o build/mesh2_f64.o sdf_mesh2_inst.hip & pids="$pids $!" ;;
    m2full) $HIPCC $FLAGS -DMESH_FULL=1 -DMESH_NAME=sdf_launch_mesh2_f64_full -c -o build/mesh2_f64_full.o sdf_mesh2_inst.hip & pids="$pids $!" ;;
    plain) $HIPCC $PLAIN -c -o build/sdf_plain.o sdf_plain.hip & pids="$pids $!" ;;
    weld) $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c -o build/sdf_weld.o sdf_weld.hip & pids="$pids $!" ;;
    *) echo "unknown unit $u"; exit 2 ;;
  esac
done
for p in $pids; do wait $p; done
exec $HIPCC --offload-arch=gfx950 -fPIC -shared -o libsdf_hip.so build/sdf_hip.o build/mesh_f64.o build/mesh_f64_full.o build/mesh2_f64.o build/mesh2_f64_full.o \
    build/sdf_bounds.o build/sdf_weld.o build/sdf_plain.o
