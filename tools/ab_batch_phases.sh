#!/bin/bash
# A/B: pre-processing of a lane-step as four launches of level-fused batch kernels (default) against one launch per kernel.
tag=${1:-bp}
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
run c2_lane1_fused "--lanes 1 --batch 96"
run c2_lane1_plain "--lanes 1 --batch 96 --no-batch-phases"
run c2_fused ""
run c2_plain "--no-batch-phases"
run c3_lane1_fused "--config 3 --lanes 1 --batch 128"
run c3_lane1_plain "--config 3 --lanes 1 --batch 128 --no-batch-phases"
run c3_fused "--config 3"
run c3_plain "--config 3 --no-batch-phases"
