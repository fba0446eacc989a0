#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
for cb in 256 128 64; do
  echo "== SDF_CULL_BLOCK=$cb"
  SDF_CULL_BLOCK=$cb timeout 600 python tools/modeltime.py --on-only gearlike:30 weave:27 weave:33 knurling:27 2>&1 | grep passes | cut -c1-100
done | tee $O/cullblock.txt
( time timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval or trig or slab or two_pass" ) > $O/t1.txt 2>&1
echo "t1 rc=$?"; tail -3 $O/t1.txt
( time SDF_CULL_BLOCK=64 timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "interval_passes or trig" ) > $O/t2.txt 2>&1
echo "t2 rc=$?"; tail -3 $O/t2.txt
