#!/bin/bash
# r06p: VERDICT r05 item 3 "explain C3": gearlike 2^30 per step on the driver's boxes went 1.349 (r03) -> 1.404 (r04) -> 1.510 (r05) and four calls in
# flight were slower than one.  The trees of round 3 and round 5 as handed in (git archive into ablibs/tree_r03, tree_r05, built with their own
# build.sh) against HEAD on ONE box, alternating: bench.py --model gearlike --samples-log2 30 with 4 in flight and with --sync, and blobby 2^30 the same.
# (the trees: for c in r03:382e653 r05:88f9b02; do mkdir -p ablibs/tree_${c%%:*}; git archive ${c##*:} | tar -x -C ablibs/tree_${c%%:*}; (cd ablibs/tree_${c%%:*}/sdf_amd/csrc && sh build.sh); done -- ablibs/ is git-ignored and travels with gpurun)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-r06p}
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for tree in r03 r05 head; do
    if [ $tree = head ]; then D=$ROOT; else D=$ROOT/ablibs/tree_$tree; fi
    for model in gearlike blobby; do
      ( cd $D && timeout 300 python bench.py --model $model --samples-log2 30 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-f32-envelope --no-check --inflight 4 > $O/${model}_${tree}_inflight4_$rep.txt 2>&1 )
      ( cd $D && timeout 300 python bench.py --model $model --samples-log2 30 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-f32-envelope --no-check --sync > $O/${model}_${tree}_sync_$rep.txt 2>&1 )
    done
  done
done
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.txt')):
    got=False
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l); got=True
            iso=r.get('isolated_calls') or {}
            print(os.path.basename(f), 'ms/step', r['ms_per_step'], 'kernel', (r.get('roofline') or {}).get('kernel_ms'), 'iso_wall', (iso.get('wall_ms') or {}).get('median') if isinstance(iso.get('wall_ms'),dict) else iso.get('wall_ms'))
    if not got: print(os.path.basename(f), 'NO LINE', open(f).read()[-300:].replace('\n',' | '))
PY
