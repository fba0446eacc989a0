#!/bin/bash
# quick HBM traffic check of k_mesh: FETCH_SIZE / WRITE_SIZE
REPO=$(pwd); OUT=$REPO/gpurun_out/pmcq; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o $c -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/$c.log 2>&1
  python - "$OUT/$c" $c <<'PY'
import sys, glob, csv
d, c = sys.argv[1], sys.argv[2]
tot = {}
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c:
            k = r['Kernel_Name'][:40]
            tot.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in tot.items():
    print(c, k, 'mean', sum(v) / len(v), 'n', len(v))
PY
done
