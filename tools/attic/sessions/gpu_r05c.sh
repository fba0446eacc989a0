#!/bin/bash
# round 5, second session: split meshing (k_sample + k_march) on a GPU for the first time.
# (1) its own tests, (2) the whole suite with the split scheme as the library's default, (3) bench lines split / one kernel,
# alternating, synchronous and pipelined, (4) kernel statistics of a synchronous run (what k_sample and k_march take).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu.py -m gpu -x -q -k "split_meshing" ) > $O/t_split.txt 2>&1
echo "split tests rc=$?"; tail -12 $O/t_split.txt | head -8
for rep in 1 2; do
  for sp in 1 0; do
    SDF_MESH_SPLIT=$sp timeout 200 python bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_sync_split${sp}_$rep.txt 2>&1
    SDF_MESH_SPLIT=$sp timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-f32-envelope > $O/bench_pipe_split${sp}_$rep.txt 2>&1
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
for sp in 1 0; do
  SDF_MESH_SPLIT=$sp timeout 300 python tools/modeltime.py --on-only example:27 pawn:27 knurling:27 blobby:30 gearlike:30 weave:27 > $O/models_split$sp.txt 2>&1
done
grep -h passes $O/models_*.txt | cut -c1-110
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/stats -o stats -- python $OLDPWD/bench.py --steps 40 --warmup 5 --sync --no-cpu-baseline --no-other-configs --no-f32-envelope > $OLDPWD/$O/stats.log 2>&1
cd $OLDPWD
python - "$O" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
( time timeout 900 python -m pytest tests/ -m gpu -x -q ) > $O/t_all.txt 2>&1
echo "suite rc=$?"; tail -5 $O/t_all.txt
