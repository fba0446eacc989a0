#!/bin/bash
# round 5: the driver's command times 20 steps: which queue depth serves a 20-step burst best? (alternating, three rounds)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05q; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do
  for d in 2 3 4 5 6 8; do
    timeout 100 python bench.py --steps 20 --warmup 5 --inflight $d --no-cpu-baseline --no-other-configs --no-check > $O/bench_d${d}_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l); res[r['steps_in_flight']].append((r['ms_per_step'], r['sustained']['ms_per_step']))
for d,v in sorted(res.items()): print('in flight', d, '20-step ms/step', [x[0] for x in v], 'sustained', [x[1] for x in v])
PY
