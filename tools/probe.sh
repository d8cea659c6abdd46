# scratch probe: tests + headline + one-lane kernel stats of config 2 (tag = $1, tests = $2)
TAG=${1:-r03z}
TESTS=${2:-"tests/test_gpu_stages.py tests/test_gpu_stream.py"}
python -m pytest $TESTS -m gpu -x -q > gpurun_out/${TAG}_test.log 2>&1; tail -3 gpurun_out/${TAG}_test.log
run() {
  python bench.py --steps 60 --warmup 10 --no-h2d --no-cpu-baseline $2 > gpurun_out/${TAG}_$1.json 2>/dev/null
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_$1.json")); r = d["roofline"]
    print("$1", d["value"], r["stage_us_per_frame_one_lane"])
except Exception as e:
    print("$1 failed", e)
PY
}
run c2_a "--config 2"
run c2_b "--config 2"
run c3 "--config 3"
run c5 "--config 5"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_c2_one_lane_plain -- python3 $GRAFT_REPO_ROOT/bench.py --config 2 --lanes 1 --batch 96 --no-cpu-baseline --no-h2d --steps 20 --warmup 2 --no-batch-phases > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/${TAG}_c2_one_lane_plain/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
