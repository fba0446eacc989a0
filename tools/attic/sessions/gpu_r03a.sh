#!/bin/bash
# round-3 GPU session A: new tests first, then the A/B measurements (spin wait, two-pass pipelined)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
rocm-smi --showclocks > $O/clocks_idle.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "rccl or skip_test_in_pieces or bench_two_ranks or slab_exchange or native_library" ) > $O/t1.txt 2>&1
echo "t1 rc=$?"; tail -5 $O/t1.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.txt 2> $O/bench_default.err
echo "bench rc=$?"; cut -c1-1500 $O/bench_default.txt
( time timeout 300 env SDF_WAIT_SPIN_US=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs ) > $O/bench_nospin.txt 2>&1
echo "nospin rc=$?"
( time timeout 300 env SDF_MESH_TWOPASS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs ) > $O/bench_twopass.txt 2>&1
echo "twopass rc=$?"
( time timeout 300 env SDF_MESH_TWOPASS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --inflight 2 ) > $O/bench_twopass2.txt 2>&1
( time timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --sync ) > $O/bench_sync.txt 2>&1
( time timeout 300 python tools/disttime.py 30 ) > $O/disttime.txt 2>&1
echo "disttime rc=$?"; tail -6 $O/disttime.txt
( time timeout 1500 python -m pytest tests/test_full_size.py -m gpu -x -q -k "exchange or c4" ) > $O/t2.txt 2>&1
echo "t2 rc=$?"; tail -5 $O/t2.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/bench_*.txt')):
    for l in open(f):
        if l.startswith('{"metric"'):
            r=json.loads(l)
            print(f.split('/')[-1], 'ms/step', r['ms_per_step'], 'lat', r['latency_ms_per_call'], 'iso', json.dumps(r['isolated_calls'])[:600], 'pipe', r['roofline']['kernel_ms_pipelined'])
PY
