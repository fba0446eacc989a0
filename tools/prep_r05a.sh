#!/bin/bash
# CPU side of the first GPU session of round 5: the libraries it compares, built here (hipcc cross-compiles) into ablibs/
#   lib_main.so     the product source of this checkout
#   lib_ldstri.so   branch next-tritab (the triangle table in LDS), built from a scratch worktree
#   + the knock-outs of tools/ablate_build.sh that were prepared but not run (noxf nolist staticchunk)
# then:  gpurun --timeout 600 -- 'bash tools/sessions/gpu_r05a.sh'
set -eu
cd "$(dirname "$0")/.."
mkdir -p ablibs
(cd sdf_amd/csrc && sh build.sh > /tmp/build_main.log 2>&1) && cp sdf_amd/csrc/libsdf_hip.so ablibs/lib_main.so && echo built main
W=/tmp/wt_tritab
rm -rf $W && git worktree prune && git worktree add -f --detach $W next-tritab > /dev/null
(cd $W/sdf_amd/csrc && sh build.sh > /tmp/build_tritab.log 2>&1) && cp $W/sdf_amd/csrc/libsdf_hip.so ablibs/lib_ldstri.so && echo built ldstri
git worktree remove --force $W
bash tools/ablate_build.sh > /tmp/ablate.log 2>&1 && echo built knock-outs
rm -f ablibs/lib_edge.so ablibs/lib_at.so ablibs/lib_atedge.so ablibs/lib_store.so ablibs/lib_tritabatedge.so ablibs/lib_tritab.so ablibs/lib_base.so
ls -la ablibs
