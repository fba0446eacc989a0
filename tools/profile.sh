#!/bin/bash
# Profile recipe for the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag> [bench args...]
# 1. rocprofv3 --kernel-trace --stats  -> per-kernel durations of the bench command
# 2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) -> HBM traffic
# 3. rocprofv3 --pmc SQ_* instruction mix of the dominant kernel
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); tools/summarize_prof.py condenses it into
# profiles/<tag>_*.{md,json}, which are the files that get committed.
set -u
TAG=${1:-r01}
shift || true
ARGS=${*:---steps 20 --warmup 3 --no-cpu-baseline}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
echo "$ARGS" > "$OUT/args.txt"
export TMPDIR=/tmp
cd /tmp
run() {   # name, rocprof args...
    local name=$1; shift
    # (the drop-in sections of the line -- generate_e2e, value_incl_d2h -- launch k_mesh in its RECORDS form, 16 bytes per triangle: left out,
    # so that every k_mesh launch of a profiled run writes the float64 soup the roofline is about)
    SDF_BENCH_SKIP=e2e,incl timeout 600 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" $ARGS \
        > "$OUT/$name.log" 2>&1
    echo "[$name] rc=$?"
    grep -h '^{"metric"' "$OUT/$name.log" | tail -1 | cut -c1-400
}
run stats --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU
run pmc_sq2 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM
cd "$REPO"
python tools/summarize_prof.py "$TAG" "$OUT" || true
find "$OUT" -name '*.csv' | head -40
