#!/bin/bash
# A/B: rows per strip of the level-0 blur inside k_blur_pyr (LM_TUNE_BLUR_STRIP)
tag=${1:-bs}
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
for rep in a b; do
  for st in 16 32 64; do run c2_s${st}_$rep "--config 2 --blur-strip $st"; done
  for st in 32 64; do run c3_s${st}_$rep "--config 3 --blur-strip $st"; done
done
