# kernel stats + vector-ALU counters of the one-lane bench with a given scan form:  bash tools/prof_scan_form.sh <outdir> <config> <form> [extra bench args]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; CFG=$2; FORM=$3; shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $CFG = 2 ]; then BL=96; elif [ $CFG = 3 ]; then BL=128; else BL=8; fi
ONE="--config $CFG --lanes 1 --batch $BL --no-cpu-baseline --no-h2d --no-pose-e2e --no-latency --steps 20 --warmup 2 --no-batch-phases --scan-form $FORM $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c${CFG}_f${FORM} -- python3 $R/bench.py $ONE > $O/c${CFG}_f${FORM}.json 2> $O/c${CFG}_f${FORM}.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_c${CFG}_f${FORM} -- python3 $R/bench.py $ONE > /dev/null 2> $O/sq_c${CFG}_f${FORM}.err
python3 - <<PY
import csv, glob, collections
ks = glob.glob("$O/ks_c${CFG}_f${FORM}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(ks)):
    if r["Name"].startswith(("k_scan", "k_refine", "k_sort", "k_merge", "k_lm_fast")) or "scan" in r["Name"]:
        print("%-50s calls %5s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
cc = glob.glob("$O/sq_c${CFG}_f${FORM}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    if not k.startswith("k_scan"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k[:40], {c: round(x / n[(k, c)] / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
