#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box into gpurun_out/<tag>_*:
#   kernel stats of the default two-lane bench and of the clean one-lane bench (configs 2 and 3),
#   FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as MI355X_MICROARCH.md prescribes), summarised per kernel.
# usage (from the repo root on the GPU box): bash tools/collect_profiles.sh r02a
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-h2d --steps 20 --warmup 2"
for CFG in 2 3; do
  if [ $CFG = 2 ]; then BL=96; else BL=128; fi     # frames per lane-launch of bench.py's default shape for the config
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_lanes2 -- python3 $R/bench.py --config $CFG $COMMON > $OUT/${TAG}_c${CFG}_lanes2.json 2> $OUT/${TAG}_c${CFG}_lanes2.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c${CFG}_lanes1 -- python3 $R/bench.py --config $CFG --lanes 1 --batch $BL $COMMON > $OUT/${TAG}_c${CFG}_lanes1.json 2> $OUT/${TAG}_c${CFG}_lanes1.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcF -- python3 $R/bench.py --config $CFG --lanes 1 --batch $BL $COMMON > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcF.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_c${CFG}_pmcW -- python3 $R/bench.py --config $CFG --lanes 1 --batch $BL $COMMON > /dev/null 2> $OUT/${TAG}_c${CFG}_pmcW.err
  python3 $R/profiles/summarize_pmc.py $OUT/${TAG}_c${CFG}_pmcF $OUT/${TAG}_c${CFG}_pmcW $BL $OUT/${TAG}_pmc_c${CFG}_batch${BL}.json $CFG > $OUT/${TAG}_pmc_c${CFG}.txt 2>&1
done
