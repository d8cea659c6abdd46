#!/bin/bash
# config 5 on one lane under rocprofv3 --kernel-trace --stats, with the knobs of r04 (usage: bash tools/prof_config5.sh <tag> "<extra bench args>" <suffix>)
TAG=${1:-r04}
EXTRA=${2:-}
SUF=${3:-default}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c5_one_lane_${SUF} -- python3 $R/bench.py --config 5 --lanes 1 --batch 8 --no-cpu-baseline --no-h2d --steps 20 --warmup 2 --no-batch-phases $EXTRA > $R/gpurun_out/${TAG}_c5_one_lane_${SUF}.json 2> $R/gpurun_out/${TAG}_c5_one_lane_${SUF}.err
cd $R
python3 - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/${TAG}_c5_one_lane_${SUF}/**/*kernel_stats.csv", recursive=True)[0]
print("== ${SUF} ${EXTRA}")
for r in list(csv.DictReader(open(f)))[:18]:
    print("%-60s %4s %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
try:
    d = json.load(open("gpurun_out/${TAG}_c5_one_lane_${SUF}.json"))
    print(d["value"], d["roofline"]["stage_us_per_frame_one_lane"], d["config"].get("list_lengths"))
except Exception as e:
    print("no json", e)
PY
