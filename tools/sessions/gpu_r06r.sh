#!/bin/bash
# r06r: is gearlike 2^30 slower INSIDE the default bench line (other_configs) than measured on its own, on the same box?
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06r}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --model gearlike --samples-log2 30 --steps 24 --warmup 2 --no-cpu-baseline --no-other-configs --no-check --inflight 4 > $O/alone_inflight4_a.txt 2>&1
timeout 300 python bench.py --model gearlike --samples-log2 30 --steps 24 --warmup 2 --no-cpu-baseline --no-other-configs --no-check --sync > $O/alone_sync_a.txt 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/default.txt 2> $O/default.err
timeout 300 python bench.py --model gearlike --samples-log2 30 --steps 24 --warmup 2 --no-cpu-baseline --no-other-configs --no-check --inflight 4 > $O/alone_inflight4_b.txt 2>&1
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(os.path.basename(f), 'ms/step', r['ms_per_step'], 'sustained', (r.get('sustained') or {}).get('ms_per_step'), 'e2e', ((r.get('generate_e2e') or {}).get('wall_ms') or {}).get('median'))
            for o in r.get('other_configs') or []:
                print('   ', o.get('workload'), o.get('ms_per_step'), o.get('ms_per_step_by_depth'), o.get('device_ms'), (o.get('whole_soup_vs_oracle') or {}))
            print('    cpu', json.dumps(r.get('cpu_baseline'))[:200])
PY
tail -4 $O/default.err
