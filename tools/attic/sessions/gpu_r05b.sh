#!/bin/bash
# round 5, first session: (1) the new circular_array axis tests + the identity tests + C2 full size under the current
# source (LDS triangle table + axis fix), (2) A/B of the round-4 library against it (bench line, alternating),
# (3) data for the next step: the two-pass scheme at C2 (how much of k_mesh is NOT sampling + counting), the phase
# counters, weave 2^33 / gearlike with both libraries (what the axis branch costs the trig tapes).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp
export SDF_HIP_LIB=$PWD/ablibs/lib_main.so   # (the committed source, built from a worktree: the tree itself is mid-edit)
( time timeout 500 python -m pytest tests/test_gpu.py -m gpu -x -q \
    -k "circular or interval or prune or cull or ragged or edge or random_csg or leaf or one_pass or tail or schemes or deferred or golden or fixture" ) > $O/t_identity.txt 2>&1
echo "identity rc=$?"; tail -3 $O/t_identity.txt | head -1
( timeout 300 python -m pytest tests/test_full_size.py -m gpu -x -q -k "matches_oracle_and_reference and (c2 or c3)" ) > $O/t_full.txt 2>&1
echo "full size rc=$?"; tail -1 $O/t_full.txt
timeout 400 bash tools/gpu_abn.sh r05b_ab r04 main
SDF_MESH_TWOPASS=1 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_twopass.txt 2>&1
SDF_MESH_TWOPASS=1 timeout 200 python bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_twopass_sync.txt 2>&1
timeout 200 python bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_sync.txt 2>&1
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'device_ms', r.get('device_ms'), 'parity', r['parity_check'])
PY
SDF_MESH_PROF=1 timeout 200 python tools/modeltime.py --on-only example:27 > $O/prof_example.txt 2>&1; grep -a "prof\]" $O/prof_example.txt | tail -8
for v in r04 main; do
  SDF_HIP_LIB=$PWD/ablibs/lib_$v.so timeout 300 python tools/modeltime.py --on-only example:27 gearlike:30 weave:27 weave:33 > $O/models_$v.txt 2>&1
done
grep -h passes $O/models_*.txt | cut -c1-110
