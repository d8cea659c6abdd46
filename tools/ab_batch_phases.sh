#!/bin/bash
# A/B: pre-processing of a lane-step as launches of level-fused batch kernels (default) against one launch per kernel.
# usage (GPU box): bash tools/ab_batch_phases.sh <tag> [configs, default "2 3"]
tag=${1:-bp}
cfgs=${2:-"2 3"}
run() { # name, args
  python bench.py --steps 60 --warmup 10 --no-h2d --no-cpu-baseline $2 > gpurun_out/${tag}_$1.json 2>/dev/null
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_$1.json")); r = d["roofline"]
    print("$1", d["value"], r["stage_us_per_frame_one_lane"])
except Exception as e:
    print("$1 failed", e)
PY
}
for c in $cfgs; do
  b=96; [ "$c" = "3" ] && b=128
  run c${c}_lane1_fused "--config $c --lanes 1 --batch $b"
  run c${c}_lane1_plain "--config $c --lanes 1 --batch $b --no-batch-phases"
  run c${c}_fused "--config $c"
  run c${c}_plain "--config $c --no-batch-phases"
done
