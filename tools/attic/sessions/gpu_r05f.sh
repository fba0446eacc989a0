#!/bin/bash
# round 5: split meshing after the counters of r05e -- k_sample without its work counter (static runs), k_march counting and emitting
# in launches of their own (nothing waits) or in one (the look-back waits), 256 / 1024 threads.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05f
mkdir -p $O
export TMPDIR=/tmp
for two in 1 0; do
( SDF_MARCH_TWO=$two timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "split_meshing" ) > $O/t_split_two$two.txt 2>&1
echo "split tests (two=$two) rc=$?"; tail -3 $O/t_split_two$two.txt | head -2
done
for rep in 1 2; do
  for v in "0 256 0" "1 256 1" "1 256 0" "1 1024 1" "1 512 1"; do
    set -- $v
    SDF_MESH_SPLIT=$1 SDF_MARCH_BLOCK=$2 SDF_MARCH_TWO=$3 timeout 200 python bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_sync_s$1_b$2_t$3_$rep.txt 2>&1
    SDF_MESH_SPLIT=$1 SDF_MARCH_BLOCK=$2 SDF_MARCH_TWO=$3 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_pipe_s$1_b$2_t$3_$rep.txt 2>&1
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.txt')):
    ok=False
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l); ok=True
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'device_ms', r.get('device_ms'), 'parity', r['parity_check'])
    if not ok: print(f.split('/')[-1], 'NO LINE'); print(open(f).read()[-1500:])
PY
for v in "1 256 1" "1 1024 1"; do
  set -- $v
  SDF_MESH_SPLIT=$1 SDF_MARCH_BLOCK=$2 SDF_MARCH_TWO=$3 timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models_s$1_b$2_t$3.txt 2>&1
  echo "== split $1 march block $2 two $3"; grep -h passes $O/models_s$1_b$2_t$3.txt | cut -c1-110
done
cd /tmp
for v in "256 1" "256 0" "1024 1"; do
set -- $v
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stats_b$1_t$2 -o stats -- env SDF_MARCH_BLOCK=$1 SDF_MARCH_TWO=$2 python $OLDPWD/bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $OLDPWD/$O/stats_b$1_t$2.log 2>&1
done
cd $OLDPWD
python - "$O" <<'PY'
import csv,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/stats*/**/*kernel_stats.csv', recursive=True)):
    print(f.split('/')[-3])
    for r in list(csv.DictReader(open(f)))[:8]:
        print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
find $O -name '*kernel_trace.csv' -size +2M -delete
