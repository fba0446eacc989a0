#!/bin/bash
# round 5: after the clean-up (split removed, float32 meshing removed, bounds kernel reworked, batch_size > 32): the whole suite, the
# bounds estimate per model, the default bench line (with generate_e2e / sustained), smoke.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/boundstime.py > $O/bounds.txt 2>&1; cat $O/bounds.txt | tail -7
( time timeout 900 python -m pytest tests/ -m gpu -x -q ) > $O/t_all.txt 2>&1
echo "suite rc=$?"; tail -6 $O/t_all.txt | head -3
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 600 python bench.py ) > $O/bench_default.txt 2> $O/bench_default.err
python - "$O" <<'PY'
import json,sys
for l in open(sys.argv[1]+'/bench_default.txt'):
    if l.startswith('{"metric"'):
        r=json.loads(l)
        print('ms/step', r['ms_per_step'], 'value', r['value'], 'lat', r['latency_ms_per_call'], 'parity', r['parity_check'])
        print('roofline', {k: r['roofline'][k] for k in ('achieved','frac','kernel_ms','traffic')}, r['roofline']['valu']['valu_issue_fraction'])
        print('e2e', r['generate_e2e'])
        print('sustained', r['sustained'])
        print('cpu', r['cpu_baseline']['value'], r['cpu_port']['value'] if r.get('cpu_port') else None)
        for o in r['other_configs'] or []: print(o.get('workload'), o.get('ms_per_step'), o.get('triangles_match_reference'), o.get('soup_sha256_equals_reference'), o.get('error'))
PY
tail -3 $O/bench_default.err
