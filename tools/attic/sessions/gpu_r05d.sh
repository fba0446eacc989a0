#!/bin/bash
# round 5, third session: split meshing reworked -- k_sample flat over arena units (a wave per run of units), k_march with its own
# sign bits and 256 / 512 / 1024 threads per workgroup.  Tests, then bench lines (synchronous + pipelined) per variant, alternating,
# then kernel statistics of the variants.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "split_meshing or batch_size_above or b64 or b48 or b128 or b40 or bounds" ) > $O/t_split.txt 2>&1
echo "split + bs tests rc=$?"; tail -12 $O/t_split.txt | head -8
for rep in 1 2; do
  for v in "0 256" "1 256" "1 512" "1 1024"; do
    set -- $v
    SDF_MESH_SPLIT=$1 SDF_MARCH_BLOCK=$2 timeout 200 python bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_sync_s$1_b$2_$rep.txt 2>&1
    SDF_MESH_SPLIT=$1 SDF_MARCH_BLOCK=$2 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_pipe_s$1_b$2_$rep.txt 2>&1
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
for v in "0 256" "1 256" "1 1024"; do
  set -- $v
  SDF_MESH_SPLIT=$1 SDF_MARCH_BLOCK=$2 timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 weave:33 > $O/models_s$1_b$2.txt 2>&1
  echo "== split $1 march block $2"; grep -h passes $O/models_s$1_b$2.txt | cut -c1-110
done
cd /tmp
for b in 256 1024; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stats$b -o stats -- env SDF_MARCH_BLOCK=$b python $OLDPWD/bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $OLDPWD/$O/stats$b.log 2>&1
done
cd $OLDPWD
python - "$O" <<'PY'
import csv,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/stats*/**/*kernel_stats.csv', recursive=True)):
    print(f.split('/')[-3])
    for r in list(csv.DictReader(open(f)))[:8]:
        print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
