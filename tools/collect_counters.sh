#!/bin/bash
# Round evidence on the GPU box: kernel stats + PMC passes of bench.py, one lane (clean kernels) and default lanes.
#   rocprofv3 --kernel-trace --stats            -> <tag>_c<cfg>_{one_lane,default_lanes,one_lane_plain}/  (per-kernel times)
#   rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE     -> HBM-side traffic per kernel (separate passes, MI355X_MICROARCH.md "HBM")
#   rocprofv3 --pmc TCP_* | TCC_*               -> L1 -> L2 requests, L2 hits / misses per kernel (scan, refine)
# Counter passes carry --kernel-trace only (no --stats / sys traces).  The program follows `--` directly.
# usage (repo root on the GPU box): bash tools/collect_counters.sh r03
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-h2d --steps 20 --warmup 2"
for CFG in 2 3; do
  if [ $CFG = 2 ]; then BL=96; else BL=128; fi
  ONE="--config $CFG --lanes 1 --batch $BL $COMMON"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_default_lanes -- python3 $R/bench.py --config $CFG $COMMON > $OUT/${TAG}_c${CFG}_default_lanes.json 2> $OUT/${TAG}_c${CFG}_default_lanes.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_one_lane -- python3 $R/bench.py $ONE > $OUT/${TAG}_c${CFG}_one_lane.json 2> $OUT/${TAG}_c${CFG}_one_lane.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_one_lane_plain -- python3 $R/bench.py $ONE --no-batch-phases > $OUT/${TAG}_c${CFG}_one_lane_plain.json 2> $OUT/${TAG}_c${CFG}_one_lane_plain.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcF -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcF.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcW -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcW.err
  python3 $R/profiles/summarize_pmc.py $OUT/${TAG}_c${CFG}_pmcF $OUT/${TAG}_c${CFG}_pmcW $BL $OUT/${TAG}_pmc_c${CFG}_batch${BL}.json $CFG > $OUT/${TAG}_pmc_c${CFG}.txt 2>&1
  # L1 -> L2 requests and L2 hits: the pruned scan (default) and the exhaustive one (calibrates bytes per request: its
  # requested bytes are known exactly)
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcTCP -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcTCP.err
  rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcTCC -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcTCC.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcTCP_noprune -- python3 $R/bench.py $ONE --no-batch-phases --no-prune > $OUT/${TAG}_c${CFG}_noprune.json 2> $OUT/${TAG}_c${CFG}_pmcTCP_noprune.err
  python3 $R/profiles/summarize_l2.py $OUT/${TAG}_c${CFG}_pmcTCP $OUT/${TAG}_c${CFG}_pmcTCC $OUT/${TAG}_c${CFG}_pmcTCP_noprune $OUT/${TAG}_c${CFG}_noprune.json $BL $CFG $OUT/${TAG}_l2_counters_c${CFG}_batch${BL}.json > $OUT/${TAG}_l2_counters_c${CFG}.txt 2>&1
done
ls $OUT | grep ${TAG}_ | head -60
