#!/bin/bash
# round 5: does k_mesh still need its park slots (1.2 GB per call lane)?  SDF_PARK=0 (a waiting batch whose predecessors are still not
# counted WAITS) against the default, alternating; steps in flight 6 / 8; the new bench fields on the default line.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05j; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  for pk in 1 0; do
    SDF_PARK=$pk timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-configs > $O/bench_park${pk}_$rep.txt 2>&1
  done
done
timeout 200 python bench.py --steps 200 --warmup 10 --inflight 8 --no-cpu-baseline --no-other-configs > $O/bench_inflight8.txt 2>&1
timeout 200 python bench.py --steps 200 --warmup 10 --inflight 4 --no-cpu-baseline --no-other-configs > $O/bench_inflight4.txt 2>&1
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'lat', r['latency_ms_per_call'], 'k_mesh', r['isolated_calls']['k_mesh_ms_hip_events']['median'], 'parity', r['parity_check'])
PY
for pk in 1 0; do
  SDF_PARK=$pk timeout 300 python tools/modeltime.py --on-only pawn:27 knurling:27 blobby:30 gearlike:30 > $O/models_park$pk.txt 2>&1
  echo "== SDF_PARK=$pk"; grep -h passes $O/models_park$pk.txt | cut -c1-100
done
( time timeout 900 python bench.py ) > $O/bench_default.txt 2> $O/bench_default.err
python - "$O" <<'PY'
import json,sys
for l in open(sys.argv[1]+'/bench_default.txt'):
    if l.startswith('{"metric"'):
        r=json.loads(l)
        print('default: ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'e2e', r['generate_e2e']['wall_ms'], r['generate_e2e']['of_which_ms']['estimate_bounds'])
        for o in r['other_configs'] or []: print(o.get('workload'), o.get('ms_per_step'), o.get('whole_soup_vs_oracle'), o.get('error'))
PY
tail -4 $O/bench_default.err
