#!/bin/bash
# r03k: final state of round 3: the whole GPU suite (with test_expand_synthetic_slabs and test_float32_envelope), the float32
# envelope at the BASELINE sizes, the multi-GPU step's stage times (warmed per configuration), the default bench line.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1
echo "tests rc=$?"; grep -a "passed\|failed\|error" $O/tests.txt | tail -3
( timeout 300 python tools/f32envelope.py ) 2>/dev/null | grep '^{' > $O/f32envelope.jsonl; cut -c1-420 $O/f32envelope.jsonl
( timeout 200 python tools/disttime.py 60 ) 2>&1 | grep "in flight" | tee $O/disttime.txt
( timeout 400 python bench.py ) > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt | cut -c1-600
