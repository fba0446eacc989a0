#!/bin/bash
# round-3 GPU session D: rotation-form circular_array -- model times, then every test that touches the trig models
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/modeltime.py --on-only example:27 gearlike:30 weave:27 weave:33 knurling:27 ) > $O/models.txt 2>&1
grep passes $O/models.txt | cut -c1-120
( time timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "values_match or interval or trig or circ or weave or gearlike or knurling or generate_matches or slab or two_pass or tail_of or float32 or weld" ) > $O/t1.txt 2>&1
echo "t1 rc=$?"; tail -4 $O/t1.txt
( time timeout 1500 python -m pytest tests/test_full_size.py -m gpu -x -q ) > $O/t2.txt 2>&1
echo "t2 rc=$?"; tail -4 $O/t2.txt
