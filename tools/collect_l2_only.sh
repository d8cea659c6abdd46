#!/bin/bash
# The L1 -> L2 / L2 counter passes of tools/collect_counters.sh alone (scan + refine rooflines), then the final bench lines.
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-h2d --steps 20 --warmup 2"
for CFG in 2 3; do
  if [ $CFG = 2 ]; then BL=96; else BL=128; fi
  ONE="--config $CFG --lanes 1 --batch $BL $COMMON"
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcTCP -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcTCP.err
  rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcTCC -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcTCC.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcTCP_noprune -- python3 $R/bench.py $ONE --no-batch-phases --no-prune > $OUT/${TAG}_c${CFG}_noprune.json 2> $OUT/${TAG}_c${CFG}_pmcTCP_noprune.err
  python3 $R/profiles/summarize_l2.py $OUT/${TAG}_c${CFG}_pmcTCP $OUT/${TAG}_c${CFG}_pmcTCC $OUT/${TAG}_c${CFG}_pmcTCP_noprune $OUT/${TAG}_c${CFG}_noprune.json $BL $CFG $OUT/${TAG}_l2_counters_c${CFG}_batch${BL}.json > $OUT/${TAG}_l2_counters_c${CFG}.txt 2>&1
  cp $OUT/${TAG}_l2_counters_c${CFG}_batch${BL}.json $R/profiles/r03_l2_counters_c${CFG}_batch${BL}.json
done
cd $R
python3 bench.py > $OUT/${TAG}_bench_config2.json 2> $OUT/${TAG}_bench_config2.err
python3 bench.py --config 3 > $OUT/${TAG}_bench_config3.json 2> $OUT/${TAG}_bench_config3.err
for f in 2 3; do python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench_config$f.json')); r=d['roofline']; h=d['config']['h2d_inclusive']; print($f, d['value'], r['stage_us_per_frame_one_lane'], r['avg_launch_us'], r['frac'], d.get('roofline_refine'), h['value'], h['h2d_GBps'], d['cpu_baseline']['value'])"; done
