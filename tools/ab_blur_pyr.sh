#!/bin/bash
# A/B: k_blur_pyr with a slot's blur / pyrDown tiles back to back (1) or dealt out evenly (2); one-lane kernel time + HBM traffic + headline
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-r04h}
cd /tmp && export TMPDIR=/tmp
for CFG in 2 3; do
  if [ $CFG = 2 ]; then BL=96; else BL=128; fi
  for V in 1 2; do
    ONE="--config $CFG --lanes 1 --batch $BL --no-cpu-baseline --no-h2d --steps 20 --warmup 2 --no-batch-phases --blur-pyr $V"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_v${V}_stats -- python3 $R/bench.py $ONE > /dev/null 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_v${V}_F -- python3 $R/bench.py $ONE > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/${TAG}_c${CFG}_v${V}_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_blur_pyr" in r["Name"]: print("config $CFG blur_pyr=$V: k_blur_pyr avg %.1f us" % (float(r["AverageNs"]) / 1e3))
f = glob.glob("$OUT/${TAG}_c${CFG}_v${V}_F/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "k_blur_pyr" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE": per[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = list(per.values())[2:]
print("   FETCH_SIZE x 2 = %.1f MB per launch = %.2f MB per frame" % (2 * sum(v) / len(v) * 1024 / 1e6, 2 * sum(v) / len(v) * 1024 / 1e6 / $BL))
PY
  done
  for V in 1 2 1 2; do python3 $R/bench.py --config $CFG --steps 60 --warmup 10 --no-h2d --no-cpu-baseline --blur-pyr $V 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $CFG blur_pyr=$V', d['value'], d['roofline']['stage_us_per_frame_one_lane'])"; done
done
