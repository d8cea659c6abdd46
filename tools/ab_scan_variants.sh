#!/bin/bash
# A/B of the scan's pruning granularity on one lane (96 frames per launch) and the default-lanes headline, configs 2 and 3.
# usage (GPU box): bash tools/ab_scan_variants.sh <tag>
tag=${1:-ab}
for v in 0 16; do
  python bench.py --steps 60 --warmup 10 --lanes 1 --batch 96 --no-h2d --no-cpu-baseline --scan-variant $v > gpurun_out/${tag}_scan_v$v.json 2>/dev/null
done
python bench.py --steps 100 --warmup 10 --no-h2d --no-cpu-baseline > gpurun_out/${tag}_c2.json 2>/dev/null
python bench.py --config 3 --steps 60 --warmup 10 --no-h2d --no-cpu-baseline > gpurun_out/${tag}_c3.json 2>/dev/null
python - <<PY
import json
for f in ("${tag}_scan_v0", "${tag}_scan_v16", "${tag}_c2", "${tag}_c3"):
    try:
        d = json.load(open("gpurun_out/%s.json" % f)); r = d["roofline"]
        print(f, d["value"], "scan us", r["avg_launch_us"], "kept", r["pruning"]["feature_loads_kept"], r["pruning"].get("lane_loads_kept"),
              "m0", d["config"]["matches_frame0"], r["stage_us_per_frame_one_lane"])
    except Exception as e:
        print(f, "failed", e)
PY
