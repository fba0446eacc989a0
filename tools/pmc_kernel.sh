#!/bin/bash
# instruction mix of every kernel of one bench run (GPU box): tools/pmc_kernel.sh [kernel-name-substring]
REPO=$(pwd); OUT=$REPO/gpurun_out/pmck; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$n -o p -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$n.log 2>&1
done
python - "$OUT" "${1:-}" <<'PY'
import sys, glob, csv
tot = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        if sys.argv[2] and sys.argv[2] not in k: continue
        tot.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in sorted(tot.items()):
    print('%-42s %-22s %14.0f' % (k, c, sum(v) / len(v)))
PY
