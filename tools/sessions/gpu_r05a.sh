#!/bin/bash
# round 5, first session (after tools/prep_r05a.sh): (1) the identity tests + the schemes-agree test under the LDS-triangle-table
# library (branch next-tritab: never run on a GPU before), (2) A/B of main vs ldstri vs the prepared knock-outs, alternating,
# (3) the instruction-cache counters of k_cull's dispatches.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
( time SDF_HIP_LIB=$PWD/ablibs/lib_ldstri.so timeout 400 python -m pytest tests/test_gpu.py -m gpu -x -q \
    -k "interval or prune or cull or ragged or edge or random_csg or arrays or leaf or texture or one_pass or tail or schemes or deferred" ) > $O/t_identity_ldstri.txt 2>&1
echo "identity (ldstri) rc=$?"; tail -3 $O/t_identity_ldstri.txt | head -1
( SDF_HIP_LIB=$PWD/ablibs/lib_ldstri.so timeout 300 python -m pytest tests/test_full_size.py -m gpu -x -q -k "matches_oracle_and_reference and (c2 or c5 or pawn)" ) > $O/t_full_ldstri.txt 2>&1
echo "full size (ldstri) rc=$?"; tail -1 $O/t_full_ldstri.txt
timeout 400 bash tools/gpu_abn.sh r05a_ab main ldstri noxf nolist staticchunk emitnone x2tape x2rows
timeout 300 bash tools/pmc_icache.sh r05a_icache > $O/icache.txt 2>&1; grep -a "k_cull" $O/icache.txt | head -20
