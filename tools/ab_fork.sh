#!/bin/bash
# A/B: fork-join pre-processing chains on three streams per lane, for batches; with 4 (default) and 8 hardware queues.
tag=${1:-fork}
run() { # name, env, args
  env $2 python bench.py --steps 60 --warmup 10 --no-h2d --no-cpu-baseline $3 > gpurun_out/${tag}_$1.json 2>/dev/null
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_$1.json")); r = d["roofline"]
    print("$1", d["value"], r["stage_us_per_frame_one_lane"])
except Exception as e:
    print("$1 failed", e)
PY
}
run base1 "A=1" "--lanes 1 --batch 96"
run fork1 "A=1" "--lanes 1 --batch 96 --fork"
run fork1q8 "GPU_MAX_HW_QUEUES=8" "--lanes 1 --batch 96 --fork"
run base3 "A=1" ""
run fork3 "A=1" "--fork"
run fork3q8 "GPU_MAX_HW_QUEUES=8" "--fork"
run base3q8 "GPU_MAX_HW_QUEUES=8" ""
run c3base "A=1" "--config 3"
run c3fork "A=1" "--config 3 --fork"
run c3forkq8 "GPU_MAX_HW_QUEUES=8" "--config 3 --fork"
