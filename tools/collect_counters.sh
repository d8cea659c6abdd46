#!/bin/bash
# Round evidence on the GPU box for configs 2, 3 and 5: kernel stats + PMC passes of bench.py on ONE lane with one launch per kernel
# (clean kernel durations), kernel stats of the default-lane run, and profiles/summarize_counters.py -> <tag>_counters_c<cfg>.json,
# the file bench.py reads (it uses it only while its `meta` -- workload, launch shape, scan variant, kernel-source hash -- equals the run's).
#   rocprofv3 --kernel-trace --stats                      per-kernel times
#   rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE               HBM-side traffic per kernel (separate passes, MI355X_MICROARCH.md "HBM")
#   rocprofv3 --pmc TCP_* | TCC_*                         L1 -> L2 requests, L2 hits / misses
#   rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE                  vector-ALU activity against the kernel's cycles
#   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_*                 (r06) LDS instructions, LDS-array cycles and bank-conflict cycles (k_scanl keeps its planes in LDS)
# Counter passes carry --kernel-trace only (no --stats / sys traces).  The program follows `--` directly.
# usage (repo root on the GPU box): bash tools/collect_counters.sh r04 ["2 3 5"]
TAG=${1:-rXX}
CFGS=${2:-"2 3 5"}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-h2d --no-pose-e2e --no-latency --steps 20 --warmup 2"
for CFG in $CFGS; do
  if [ $CFG = 2 ]; then BL=96; elif [ $CFG = 3 ]; then BL=128; else BL=8; fi
  ONE="--config $CFG --lanes 1 --batch $BL $COMMON --no-batch-phases"
  P=$OUT/${TAG}_c${CFG}
  rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_default_lanes -- python3 $R/bench.py --config $CFG $COMMON > ${P}_default_lanes.json 2> ${P}_default_lanes.err
  rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_one_lane -- python3 $R/bench.py $ONE > ${P}_one_lane.json 2> ${P}_one_lane.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${P}_pmcF -- python3 $R/bench.py $ONE > /dev/null 2> ${P}_pmcF.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${P}_pmcW -- python3 $R/bench.py $ONE > /dev/null 2> ${P}_pmcW.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d ${P}_pmcTCP -- python3 $R/bench.py $ONE > /dev/null 2> ${P}_pmcTCP.err
  rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d ${P}_pmcTCC -- python3 $R/bench.py $ONE > /dev/null 2> ${P}_pmcTCC.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d ${P}_pmcSQ -- python3 $R/bench.py $ONE > /dev/null 2> ${P}_pmcSQ.err
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d ${P}_pmcLDS -- python3 $R/bench.py $ONE > /dev/null 2> ${P}_pmcLDS.err
  # the exhaustive scan calibrates bytes per L1 -> L2 request: its requested bytes are known exactly
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d ${P}_pmcTCP_noprune -- python3 $R/bench.py $ONE --no-prune > ${P}_noprune.json 2> ${P}_pmcTCP_noprune.err
  python3 $R/profiles/summarize_counters.py $OUT/${TAG}_counters_c${CFG}.json ${P}_one_lane.json ${P}_one_lane \
      F=${P}_pmcF W=${P}_pmcW TCP=${P}_pmcTCP TCC=${P}_pmcTCC SQ=${P}_pmcSQ LDS=${P}_pmcLDS TCPNP=${P}_pmcTCP_noprune NPJSON=${P}_noprune.json > $OUT/${TAG}_counters_c${CFG}.txt 2>&1
  # what gets committed: the summary, the kernel-stats tables, the two bench lines
  cp $(find ${P}_one_lane -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_config${CFG}_one_lane_kernel_stats.csv
  cp $(find ${P}_default_lanes -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_config${CFG}_default_lanes_kernel_stats.csv
  cat $OUT/${TAG}_counters_c${CFG}.txt
done
ls $OUT | grep ${TAG}_ | head -80
