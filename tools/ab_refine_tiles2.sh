#!/bin/bash
# NOTE: the LM_EXPERIMENT switch these runs need lived in lm_kernels.hip for commit 'k_refine on a tiled spread memory: measured ...' only (git log); results: profiles/r05_ab_experiments.log section 2.
# second half of tools/ab_refine_tiles.sh: the producer (k_lm_spread5 as its own launch: --no-batch-phases), 96-frame launches of config 2
set -u
OUT=${1:-gpurun_out/r05_refine_tiles}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ex in 0 4; do
  LM_EXPERIMENT=$ex rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/p2_ex${ex}" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config 2 --lanes 1 --batch 96 --no-batch-phases --steps 20 --warmup 3 --no-cpu-baseline --no-pose-e2e --no-h2d > "$GRAFT_REPO_ROOT/$OUT/p2_ex${ex}.json" 2> "$GRAFT_REPO_ROOT/$OUT/p2_ex${ex}.err"
done
for ex in 2 3; do
  LM_EXPERIMENT=$ex rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/q2_ex${ex}" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config 2 --lanes 1 --batch 96 --steps 20 --warmup 3 --no-cpu-baseline --no-pose-e2e --no-h2d > "$GRAFT_REPO_ROOT/$OUT/q2_ex${ex}.json" 2> "$GRAFT_REPO_ROOT/$OUT/q2_ex${ex}.err"
done
cd "$GRAFT_REPO_ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for tag in ("p2_ex0", "p2_ex4", "q2_ex2", "q2_ex3"):
    f = glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    print(tag, "  ".join("%s %.1f us x%s" % (r["Name"].replace("void (anonymous namespace)::", "")[:22], float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows if any(k in r["Name"] for k in ("k_lm_spread5", "k_refine<", "k_lm_fast"))))
PY
