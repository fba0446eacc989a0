#!/bin/bash
# r06ah: the headline variant with THREE / FOUR samples per lane (the interpreter's scalar stream per sample: - 33 % / - 50 %; 54 / 204 spilled registers;
# 48 / 64 tasks per sampling pass instead of 32): SDF_MESH_SHAPE=3 / 4 against the default, one box, alternating
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06ah}
mkdir -p $O
export TMPDIR=/tmp
for sh in 3 4; do SDF_MESH_SHAPE=$sh timeout 600 python -m pytest tests/test_gpu.py -m gpu -x -q -k "generate_matches and example or records and example or sparse_tiles" 2>&1 | tail -2; done | tee $O/tests.txt
for rep in 1 2 3; do for sh in 0 3 4; do SDF_MESH_SHAPE=$sh timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_shape${sh}_$rep.txt 2>&1; done; done
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(os.path.basename(f), 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'parity', r['parity_check'])
PY
for sh in 0 3 4; do SDF_MESH_SHAPE=$sh timeout 300 python tools/modeltime.py --on-only example:27 example:24 > $O/models_shape$sh.txt 2>&1; done; grep -H passes $O/models_*.txt | cut -c1-150
