#!/bin/bash
# where a batch's post-processing spends its wall time: the e2e child once more with LM_POST_TRACE=1 (stderr lines of HighLevelLineMOD::detectTemplatesBatchEnd)
OUT=${1:-gpurun_out/r05_post_trace}
mkdir -p "$OUT"
LM_POSE_E2E_KEEP=/tmp/e2e python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-latency > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cd /tmp/e2e
LM_POST_TRACE=1 ./pose_e2e_bench bench.bank poses.bin frames.raw 1280 960 8 80 6 0 16 > "$OLDPWD/$OUT/run.json" 2> "$OLDPWD/$OUT/trace.log"
cd "$OLDPWD"
grep -c post-trace "$OUT/trace.log"; tail -24 "$OUT/trace.log"
