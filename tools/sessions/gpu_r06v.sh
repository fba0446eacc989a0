#!/bin/bash
# r06v: the slow and the fast case of gearlike-with-calls-in-flight inside the bench line, on the device's clock
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06v}
mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" SDF_BENCH_WHOLE_SOUP_S=1 SDF_BENCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check > $O/$tag.txt 2> $O/$tag.err; }
run all_gear_only SDF_BENCH_OTHER_ORDER=gearlike
run skip_sustained SDF_BENCH_OTHER_ORDER=gearlike SDF_BENCH_SKIP=sustained
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            for o in r.get('other_configs') or []:
                print(os.path.basename(f), o.get('ms_per_step_by_depth'), json.dumps(o.get('in_flight_run_on_the_device_clock'))[:700])
PY
grep -n "gearlike\|submitted call\|collected" $O/all_gear_only.err | awk '/measure gearlike 2\^30: 24 steps, 4/{p=1} /measure gearlike done/{if(p)print; p=0} p' | tail -60 | cut -c1-120
