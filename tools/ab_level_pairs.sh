#!/bin/bash
# A/B: slot-interleaved level pairs (k_pair, LM_TUNE_LEVEL_PAIRS) against the plain / register-class-fused sequences; each shape twice, interleaved.
tag=${1:-lp}
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
for c in 2 3; do
  run c${c}_pairs_$rep "--config $c"
  run c${c}_nopairs_$rep "--config $c --no-level-pairs"
done
done
if true; then
  run c2_lane1_pairs "--config 2 --lanes 1 --batch 96"
  run c2_lane1_plain "--config 2 --lanes 1 --batch 96 --no-level-pairs --no-batch-phases"
  run c2_lane1_fused "--config 2 --lanes 1 --batch 96 --no-level-pairs"
  run c3_lane1_pairs "--config 3 --lanes 1 --batch 128"
  run c3_lane1_plain "--config 3 --lanes 1 --batch 128 --no-level-pairs --no-batch-phases"
  run c3_lane1_fused "--config 3 --lanes 1 --batch 128 --no-level-pairs"
fi
