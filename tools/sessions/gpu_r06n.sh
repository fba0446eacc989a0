#!/bin/bash
# r06n: the host expansion alone (records all present): scaling with threads
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06n}
mkdir -p $O
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-host-only -O3 -std=c++17 -w -ffp-contract=off -pthread -I ../../sdf_amd/csrc -o /tmp/host_expand_rate host_expand_rate.hip && timeout 300 /tmp/host_expand_rate ) > $O/host_expand_rate.txt 2>&1
cat $O/host_expand_rate.txt
