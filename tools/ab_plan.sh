#!/bin/bash
# NOTE: LM_PLAN_SNAKE existed for this A/B only (profiles/r05_ab_experiments.log section 3); the fused sort + merge it also timed was reverted.
# r05: the refine plan dealt sequentially (least loaded list first) against snake order of the ranked pieces; and the fused sort + merge.
OUT=${1:-gpurun_out/r05_plan}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_match.py -m gpu -x -q -k "sort or crowded or flood or refine" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in 2 5; do
  if [ $cfg = 2 ]; then BL=96; else BL=8; fi
  for sn in 0 1; do
    LM_PLAN_SNAKE=$sn rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/c${cfg}_snake${sn} -- python3 $R/bench.py --config $cfg --lanes 1 --batch $BL --steps 30 --warmup 3 --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e > /dev/null 2>&1
  done
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, os
for cfg in (2, 5):
    for sn in (0, 1):
        f = glob.glob(os.path.join(sys.argv[1], "c%d_snake%d" % (cfg, sn), "**", "*kernel_stats.csv"), recursive=True)
        rows = list(csv.DictReader(open(f[0])))
        print("config %d snake %d: " % (cfg, sn) + "  ".join("%s %.1f us x%s" % (r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:14], float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows if any(k in r["Name"] for k in ("k_refine", "k_sort", "k_merge"))))
PY
for sn in 0 1; do LM_PLAN_SNAKE=$sn python3 bench.py --no-cpu-baseline --no-h2d --no-latency 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 2 three lanes, snake $sn:', d['value'])"; done
for sn in 0 1; do LM_PLAN_SNAKE=$sn python3 bench.py --config 5 --no-cpu-baseline --no-h2d --no-latency --no-pose-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 5 three lanes, snake $sn:', d['value'])"; done
