#!/bin/bash
# r05aa: the default bench line of the source handed in, twice (its roofline.traffic now quotes profiles/r05y_sync_pmc.json by source_id;
# does blobby 2^30 with four calls in flight come out at 0.88 ms per step as in every other run?), f.save end to end
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05aa; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  ( timeout 600 python bench.py ) > $O/bench_$rep.txt 2> $O/bench_$rep.err; echo "bench rc=$?"
done
( timeout 300 python tools/savetime.py ) > $O/savetime.txt 2>&1; tail -12 $O/savetime.txt
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'sustained', r['sustained']['ms_per_step'], 'traffic', r['roofline']['traffic_over_algorithmic'], 'e2e', r['generate_e2e']['wall_ms']['median'], 'cpu', r['cpu_baseline'].get('value'), [(o['workload'][:8], o['steps_in_flight'], o['ms_per_step_by_depth']) for o in r['other_configs']])
PY
