#!/bin/bash
# r06x: with eight warm-up steps in other_configs: is the slow case of the first configuration gone?  (three default-like runs)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06x}
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do SDF_BENCH_OTHER_ORDER=gearlike,blobby SDF_BENCH_WHOLE_SOUP_S=1 SDF_BENCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check > $O/run$i.txt 2> $O/run$i.err; done
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(os.path.basename(f), r['ms_per_step'], [(o.get('workload','')[:8], o.get('ms_per_step_by_depth'), (o.get('in_flight_run_on_the_device_clock') or {}).get('start_to_start_ms')) for o in r.get('other_configs') or []])
PY
