#!/bin/bash
# r05: kernel + memory-copy + HIP-API timeline of the streamed PoseDetection (tools/pose_e2e_bench.cpp), to see what the lane waits for.
set -u
OUT=${1:-gpurun_out/r05_e2e_trace}
mkdir -p "$OUT"
LM_POSE_E2E_KEEP=/tmp/e2e python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err" || python bench.py --config 5 --steps 10 --warmup 3 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
ROOT=$PWD
cd /tmp/e2e && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d "$ROOT/$OUT/trace" -- ./pose_e2e_bench bench.bank poses.bin frames.raw 1280 960 8 80 6 0 16 > "$ROOT/$OUT/run.log" 2>&1
cd "$ROOT"
find "$OUT/trace" -name "*.csv" | head
