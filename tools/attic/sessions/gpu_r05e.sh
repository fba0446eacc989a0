#!/bin/bash
# round 5: why are k_sample (flat) and k_march slow?  Counters of one synchronous bench run with the split scheme.
set -u
cd "$(dirname "$0")/../.."
REPO=$PWD; O=$REPO/gpurun_out/r05e; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*\|TCP_[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_IFETCH[A-Z_]*" | sort -u | tr '\n' ' ' > $O/counters.txt
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -c1-16 | tr ' ' '_')
  SDF_MESH_SPLIT=1 timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$n -o p -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-f32-envelope --sync > $O/$n.log 2>&1
  echo "[$n] rc=$?"
done
python - "$O" <<'PY'
import sys, glob, csv
tot = {}
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-30:]
        if not ('k_sample' in k or 'k_march' in k or 'k_cull' in k): continue
        tot.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, c), v in sorted(tot.items()):
    print('%-32s %-26s %16.0f  n=%d' % (k, c, sum(v) / len(v), len(v)))
PY
find $O -name '*kernel_trace.csv' -size +2M -delete
find $O -name '*counter_collection.csv' -size +8M -delete
