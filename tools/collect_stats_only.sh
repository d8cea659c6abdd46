#!/bin/bash
# Kernel stats + final bench lines only (the counter passes of tools/collect_counters.sh are unaffected by a pre-processing change).
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-h2d --steps 20 --warmup 2"
for CFG in 2 3; do
  if [ $CFG = 2 ]; then BL=96; else BL=128; fi
  ONE="--config $CFG --lanes 1 --batch $BL $COMMON"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_default_lanes -- python3 $R/bench.py --config $CFG $COMMON > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_one_lane -- python3 $R/bench.py $ONE > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_one_lane_plain -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcF -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcW -- python3 $R/bench.py $ONE --no-batch-phases > /dev/null 2>&1
  python3 $R/profiles/summarize_pmc.py $OUT/${TAG}_c${CFG}_pmcF $OUT/${TAG}_c${CFG}_pmcW $BL $OUT/${TAG}_pmc_c${CFG}_batch${BL}.json $CFG > $OUT/${TAG}_pmc_c${CFG}.txt 2>&1
done
cd $R
python3 bench.py > $OUT/${TAG}_bench_config2.json 2> $OUT/${TAG}_bench_config2.err
python3 bench.py --config 3 > $OUT/${TAG}_bench_config3.json 2> $OUT/${TAG}_bench_config3.err
python3 bench.py --config 5 --steps 100 --warmup 10 > $OUT/${TAG}_bench_config5.json 2> $OUT/${TAG}_bench_config5.err
for f in 2 3 5; do python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench_config$f.json')); r=d['roofline']; h=d['config']['h2d_inclusive']; print($f, d['value'], r['stage_us_per_frame_one_lane'], r['avg_launch_us'], r['frac'], h['value'], h['h2d_GBps'], d['cpu_baseline']['value'])"; done
