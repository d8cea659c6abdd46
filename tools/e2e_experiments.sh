#!/bin/bash
# r05: where does the streamed PoseDetection wait?  bench.py --config 5 leaves its child's inputs in /tmp/e2e; the child is then run by hand with
# different pool sizes and with a sleep between End(k) and Begin(k + 2) (is the GPU done when nobody asks?).
set -u
OUT=${1:-gpurun_out/r05_e2e}
mkdir -p "$OUT"
LM_POSE_E2E_KEEP=/tmp/e2e python bench.py --config 5 --steps 20 --warmup 5 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cd /tmp/e2e
for t in 4 8 12 16 24 32; do
  echo "threads $t" >> "$OLDPWD/$OUT/runs.log"
  ./pose_e2e_bench bench.bank poses.bin frames.raw 1280 960 8 80 20 0 $t >> "$OLDPWD/$OUT/runs.log" 2>&1
done
for s in 1000 3000; do
  echo "threads 16 sleep $s" >> "$OLDPWD/$OUT/runs.log"
  LM_E2E_SLEEP_US=$s ./pose_e2e_bench bench.bank poses.bin frames.raw 1280 960 8 80 20 0 16 >> "$OLDPWD/$OUT/runs.log" 2>&1
done
cd "$OLDPWD"
python - "$OUT/runs.log" <<'PY'
import json, sys
tag = None
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("threads"):
        tag = l
    elif l.startswith("{"):
        d = json.loads(l)
        print(tag, "| identical", d["poses_identical_across_passes"], "|", " | ".join("%s %.0f (begin %.0f wait %.0f post %.0f)" % (k, d[k]["us_per_frame"], d[k]["in_begin_us_per_frame"], d[k]["waiting_for_the_gpu_us_per_frame"], d[k]["post_us_per_frame"]) for k in ("serial", "pipelined", "pipelined_pinned")))
PY
