#!/bin/bash
# r06m: where should the ndarray of the host expansion live?  (tools/ubench/host_write_bw.hip)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06m}
mkdir -p $O
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -pthread -Wno-unused-result -o /tmp/host_write_bw host_write_bw.hip && timeout 300 /tmp/host_write_bw ) > $O/host_write_bw.txt 2>&1
cat $O/host_write_bw.txt
numactl -H 2>/dev/null | head -20 >> $O/host_write_bw.txt; cat /sys/kernel/mm/transparent_hugepage/enabled >> $O/host_write_bw.txt; lscpu | grep -i "numa\|model name\|socket" >> $O/host_write_bw.txt
tail -12 $O/host_write_bw.txt
