#!/bin/bash
# r06u: tools/c3_phase.py -- does gearlike's overlap depend on which call slots its calls land on?
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06u}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/c3_phase.py > $O/phase.txt 2>&1
cat $O/phase.txt | tail -14
