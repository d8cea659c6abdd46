#!/bin/bash
# quick look at config 2: stage tests of the depth modality, one-lane kernel stats, the default bench line
OUT=${1:-gpurun_out/quick_c2}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "depth or float_tail" > $OUT/stages.log 2>&1; tail -2 $OUT/stages.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/one_lane -- python3 $R/bench.py --config 2 --lanes 1 --batch 96 --steps 20 --warmup 2 --no-cpu-baseline --no-h2d --no-latency > /dev/null 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "one_lane", "**", "*kernel_stats.csv"), recursive=True)
for r in list(csv.DictReader(open(f[0])))[:16]:
    print("%-44s %8.1f us x %s" % (r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
python3 bench.py --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 -c "
import json; d=json.load(open('$OUT/bench_c2.json')); print('value', d['value'], 'ms/step', d['ms_per_step'], 'h2d', d['config']['h2d_inclusive']['value']); print(json.dumps(d['latency'])[:900]); print(json.dumps(d['roofline_pipeline'])[:300])"
