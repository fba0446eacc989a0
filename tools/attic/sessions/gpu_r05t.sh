#!/bin/bash
# round 5: two interval levels in k_cull against three (the default for lean tapes) at C2 with today's k_mesh, alternating
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05t; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  for lv in 3 2; do
    SDF_CULL_LEVELS=$lv timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_lv${lv}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'prepass', r['isolated_calls']['prepass_ms']['median'], 'parity', r['parity_check'])
PY
