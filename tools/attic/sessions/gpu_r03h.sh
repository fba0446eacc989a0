#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03h
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/modeltime.py --on-only weave:27 weave:33 ) 2>&1 | grep passes | cut -c1-110
echo "== one pass"
( SDF_MESH_TWOPASS=0 timeout 600 python tools/modeltime.py --on-only weave:27 weave:33 ) 2>&1 | grep passes | cut -c1-110
( time timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "failed_alloc or two_pass or slab or rccl or sharded or tail_of" ) > $O/t1.txt 2>&1
echo "t1 rc=$?"; tail -3 $O/t1.txt
( time timeout 1500 python -m pytest tests/test_full_size.py -m gpu -x -q -k "c4 or exchange" ) > $O/t2.txt 2>&1
echo "t2 rc=$?"; tail -3 $O/t2.txt
