#!/bin/bash
# r06h: second sheet of instruction costs (selects, compares, lane moves, LDS)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06h}
mkdir -p $O
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/valu_rates2 valu_rates2.hip && timeout 300 /tmp/valu_rates2 ) > $O/valu_rates2.txt 2>&1
grep "per element" $O/valu_rates2.txt
