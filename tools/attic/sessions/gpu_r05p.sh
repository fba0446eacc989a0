#!/bin/bash
# round 5, last check of the source handed in: whole suite, smoke, the bench line as the driver runs it
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05p; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.txt 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.txt | cut -c1-300; tail -3 $O/bench.err
python - "$O" <<'PY'
import json,sys
for l in open(sys.argv[1]+'/bench.txt'):
    if l.startswith('{"metric"'):
        r=json.loads(l)
        print('roofline', {k: r['roofline'][k] for k in ('achieved','frac','kernel_ms','traffic','traffic_over_algorithmic')}, r['roofline']['valu']['valu_issue_fraction'])
        print('sustained', r['sustained']['ms_per_step'], 'e2e', r['generate_e2e']['wall_ms']['median'], 'cpu', r['cpu_baseline']['kind'][:40], r['cpu_port']['value'])
PY
