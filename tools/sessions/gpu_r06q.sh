#!/bin/bash
# r06q: `python bench.py --gpus 8` by EIGHT processes on one GPU through the stand-in for librccl (tests/native/mock_rccl.cpp):
# the whole N = 8 flow -- headline config + C3, C4 at 2^33, C5 through the library's native exchange with 16-byte slab records
# -- for its CONTENT (hashes, counts); its times mean nothing (every all-gather goes through host memory, eight processes
# share one device)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r06q
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -shared -fPIC -O2 -o /tmp/mock_rccl.so tests/native/mock_rccl.cpp || exit 1
( time env SDF_BENCH_ONE_DEVICE=1 SDF_BENCH_BACKEND=gloo SDF_BENCH_COMM_DEVICE=cuda SDF_DIST_NATIVE=force SDF_RCCL_LIB=/tmp/mock_rccl.so \
    MOCK_RCCL_SLOT_MB=512 SDF_BENCH_OTHER_TIMEOUT_S=330 SDF_BENCH_HEADLINE_TIMEOUT_S=200 \
    timeout 560 python bench.py --gpus 8 --steps 4 --warmup 1 ) > $O/bench8.txt 2> $O/bench8.err
echo "rc=$?"; tail -1 $O/bench8.txt | cut -c1-600; tail -3 $O/bench8.err
