#!/bin/bash
# A/B: cv::pyrDown as one lane per 8 output pixels (1) against the row-walking kernel with shared column sums (2).
tag=${1:-pd}
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
  run c${c}_v1 "--config $c --no-batch-phases --pyrdown-variant 1"
  run c${c}_v2 "--config $c --no-batch-phases --pyrdown-variant 2"
done
