#!/bin/bash
# r06ai: three samples per lane for the (1,1) and (2,2) register files of both families: which models gain?  SDF_MESH_SHAPE=3 against the default,
# one box, alternating; identity tests under the new shape
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06ai}
mkdir -p $O
export TMPDIR=/tmp
SDF_MESH_SHAPE=3 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "generate_matches or records or sparse_tiles or interval_passes or full_size_configs" 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2 3; do for sh in 0 3; do SDF_MESH_SHAPE=$sh timeout 300 python tools/modeltime.py --on-only example:22 example:24 example:27 example:30 gearlike:27 gearlike:30 blobby:27 blobby:30 knurling:27 > $O/models_shape${sh}_$rep.txt 2>&1; done; done
grep -H passes $O/models_*.txt | cut -c1-150 | sort -k2,3 -s
