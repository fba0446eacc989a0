#!/bin/bash
# Register / scratch / spill table of every k_mesh variant of one (precision, FULL) family, from the compiler's
# own remarks (-Rpass-analysis=kernel-resource-usage); cross-compiles, no GPU needed:
#   tools/kernel_resources.sh double 1        (T = double|float, FULL = 0|1)   [extra hipcc flags...]
T=${1:-double}; FULL=${2:-0}; shift; shift
cd "$(dirname "$0")/../sdf_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result \
    -mllvm -structurizecfg-skip-uniform-regions=1 -DMESH_T=$T -DMESH_FULL=$FULL -DMESH_NAME=probe \
    -Rpass-analysis=kernel-resource-usage -c -o /dev/null sdf_mesh_inst.hip "$@" 2>&1 | grep remark |
  sed 's/.*remark: *//; s/ \[-Rpass.*//' |
  awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' |
  sed 's/Function Name: _ZN4sdfk6k_meshI/k_mesh</; s/EEvPKjPKT_NS_8MeshArgsE/>/; s/ELi/,/g; s/Lb/,/; s/AGPRs: 0 | //; s/Dynamic Stack: False | //; s/ | LDS Size.*//'
