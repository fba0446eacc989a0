#!/bin/bash
# r06t: which part of the default bench line makes gearlike's four calls in flight lose their overlap?  (bisect by sections / order)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06t}
mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" SDF_BENCH_WHOLE_SOUP_S=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check > $O/$tag.txt 2> $O/$tag.err; }
run all_gear_only SDF_BENCH_OTHER_ORDER=gearlike
run skip_all SDF_BENCH_OTHER_ORDER=gearlike SDF_BENCH_SKIP=incl,e2e,sustained
run skip_e2e SDF_BENCH_OTHER_ORDER=gearlike SDF_BENCH_SKIP=e2e
run skip_sustained SDF_BENCH_OTHER_ORDER=gearlike SDF_BENCH_SKIP=sustained
run skip_incl SDF_BENCH_OTHER_ORDER=gearlike SDF_BENCH_SKIP=incl
run blobby_first SDF_BENCH_OTHER_ORDER=blobby,gearlike
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(os.path.basename(f), 'ms/step', r['ms_per_step'], [(o.get('workload','')[:8], o.get('ms_per_step_by_depth')) for o in r.get('other_configs') or []])
PY
