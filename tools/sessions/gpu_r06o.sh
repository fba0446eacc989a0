#!/bin/bash
# r06o: the whole GPU suite on the build with the records mode (sdf_generate_records / sdf_mesh_emit_host_workers)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r06o}
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/suite.txt
cat $O/suite.txt
