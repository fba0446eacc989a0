#!/bin/bash
# instruction-cache behaviour + HBM traffic of every kernel of one synchronous bench run (GPU box): tools/pmc_icache.sh <tag> [env...]
REPO=$(pwd); TAG=${1:-pmci}; shift || true
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | cut -c1-14 | tr ' ' '_')
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -o p -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --sync > $OUT/$n.log 2>&1
  echo "[$n] rc=$?"
done
python - "$OUT" <<'PY'
import sys, glob, csv
tot = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-44:]
        if not ('k_mesh' in k or 'k_cull' in k or 'k_skip' in k): continue
        tot.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in sorted(tot.items()):
    print('%-46s %-28s %16.0f  n=%d' % (k, c, sum(v) / len(v), len(v)))
PY
find $OUT -name '*kernel_trace.csv' -size +4M -delete
