#!/bin/bash
# Knock-out builds of the library for phase attribution under REAL overlap (the cycle counters of SDF_MESH_PROF see one
# thread; waves overlap): a scratch copy of sdf_amd/csrc gets tools/ablation_knockouts.patch (emission pieces removed behind
# -DSDF_ABL_*: results are WRONG on purpose, triangle counts and control flow unchanged), one library per variant lands in
# ablibs/lib_<name>.so for tools/gpu_abn.sh.  Nothing of this touches the product source.
#   tools/ablate_build.sh            # base edge at atedge store tritab tritabatedge emitnone (run in r04ai / r04aj) + noxf nolist staticchunk x2tape x2rows (prepared, not run)
set -eu
cd "$(dirname "$0")/.."
S=/tmp/abl
rm -rf $S && mkdir -p $S/sdf_amd && cp -r sdf_amd/csrc $S/sdf_amd/csrc && cp -r include $S/include
rm -rf $S/sdf_amd/csrc/build $S/sdf_amd/csrc/*.so
patch -s $S/sdf_amd/csrc/sdf_device.h tools/ablation_knockouts.patch
mkdir -p ablibs
build() { n=$1; shift; (cd $S/sdf_amd/csrc && sh build.sh "$@" > $S/build_$n.log 2>&1); cp $S/sdf_amd/csrc/libsdf_hip.so ablibs/lib_$n.so; echo built $n; }
build base
build edge -DSDF_ABL_EDGE                 # vertex placement without its float64 division
build at -DSDF_ABL_AT                     # ... without the two sample look-ups (TileView::at)
build atedge -DSDF_ABL_AT -DSDF_ABL_EDGE
build store -DSDF_ABL_STORE               # everything computed and staged, no soup store
build tritab -DSDF_ABL_TRITAB             # the triangle's three edges by arithmetic instead of three byte loads from the table in device memory
build tritabatedge -DSDF_ABL_TRITAB -DSDF_ABL_AT -DSDF_ABL_EDGE
build emitnone -DSDF_ABL_EMITNONE         # the emission loop's body removed (chunk hand-out and barriers stay)
# prepared for the next round (not yet run on a GPU): what is left of the emission's other half
build noxf -DSDF_ABL_NOXF                 # soup stores without the float64 scale + offset
build nolist -DSDF_ABL_NOLIST             # the triangle's list entry by arithmetic instead of the LDS read
build staticchunk -DSDF_ABL_STATICCHUNK   # a waiting batch's triangles by fixed assignment instead of the atomic hand-out of chunks (a real alternative, results stay right)
# doubling instead of removing (results stay RIGHT; the extra time is the piece's cost under real overlap):
build x2tape -DSDF_ABL_X2_TAPE            # every interpreter pass runs twice
build x2rows -DSDF_ABL_X2_ROWS            # the per-row surface masks + their block scans (count phase, 2a) run twice
