#!/bin/bash
# A/B: level-0 / level-1 Gaussian blur as k_cblur_sh (0 = default: inside k_blur_pyr) or on the matrix cores (4: k_cblur_mx + k_pyrdown16 launched apart)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-r04n}
cd /tmp && export TMPDIR=/tmp
for CFG in 2 3; do
  if [ $CFG = 2 ]; then BL=96; else BL=128; fi
  for V in 0 4; do
    ONE="--config $CFG --lanes 1 --batch $BL --no-cpu-baseline --no-h2d --steps 20 --warmup 2 --no-batch-phases --cblur-variant $V"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_v${V}_stats -- python3 $R/bench.py $ONE > /dev/null 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob("$OUT/${TAG}_c${CFG}_v${V}_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_blur_pyr", "k_blur_mx", "k_cblur", "k_pyrdown")) and int(r["Calls"]) > 8: print("config $CFG cblur=$V: %-40s %4s calls avg %.1f us" % (r["Name"].split("::")[-1][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
  for V in 0 4 0 4; do python3 $R/bench.py --config $CFG --steps 60 --warmup 10 --no-h2d --no-cpu-baseline --cblur-variant $V 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $CFG cblur=$V', d['value'], d['roofline']['stage_us_per_frame_one_lane'])"; done
done
