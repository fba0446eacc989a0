#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/opcost.py f64 sphere_x2 box translate_x4 rotate_x4 example circ_array twist rbox rbox_t rbox_tb rbox_tbc rbox_tbcr > $O/opcost.txt 2>&1
cat $O/opcost.txt
