#!/bin/bash
# A/B: sliding-window blur (2) against the form with column sums shared between lanes (3); one launch per kernel so that the
# blur's own time shows in the one-lane pre-processing figure.
tag=${1:-cb}
run() {
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
for c in 2 3; do
  run c${c}_v2 "--config $c --no-batch-phases --cblur-variant 2"
  run c${c}_v3 "--config $c --no-batch-phases --cblur-variant 3"
done
