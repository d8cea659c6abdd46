#!/bin/bash
# NOTE: the LM_EXPERIMENT switch these runs need lived in lm_kernels.hip for commit 'k_refine on a tiled spread memory: measured ...' only (git log); results: profiles/r05_ab_experiments.log section 2.
# r05: what could the 8 x 16 tiled spread memory buy?  Timing experiments with WRONG results (LM_EXPERIMENT, lm_kernels.hip):
#   2     k_refine without pruning, linear layout            (baseline of the consumer experiment: same work for both layouts)
#   3     k_refine without pruning, patch read as if tiled   (about 4 lines per patch instead of 16-17; none of the extra address arithmetic)
#   4     k_lm_spread5 storing its 8-byte pieces tile-wise   (the producer's side)
# one lane, config 2 and config 5; kernel stats by rocprofv3.
set -u
OUT=${1:-gpurun_out/r05_refine_tiles}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in 2 5; do
  for ex in 0 2 3 4; do
    LM_EXPERIMENT=$ex rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/c${cfg}_ex${ex}" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $cfg --lanes 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pose-e2e --no-h2d > "$GRAFT_REPO_ROOT/$OUT/c${cfg}_ex${ex}.json" 2> "$GRAFT_REPO_ROOT/$OUT/c${cfg}_ex${ex}.err"
  done
done
cd "$GRAFT_REPO_ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for cfg in (2, 5):
    for ex in (0, 2, 3, 4):
        f = glob.glob(os.path.join(out, "c%d_ex%d" % (cfg, ex), "**", "*kernel_stats.csv"), recursive=True)
        if not f:
            print(cfg, ex, "no stats"); continue
        rows = list(csv.DictReader(open(f[0])))
        pick = {}
        for r in rows:
            n = r["Name"]
            for key in ("k_refine<", "k_lm_spread5", "k_refine_plan", "k_scan4"):
                if key in n:
                    pick[key] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
        print("config %d LM_EXPERIMENT=%d  " % (cfg, ex) + "  ".join("%s %.1f us x%d" % (k, v[0], v[1]) for k, v in sorted(pick.items())))
PY
