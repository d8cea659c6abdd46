R=$GRAFT_REPO_ROOT; TAG=$1; cd /tmp && export TMPDIR=/tmp
for C in 2 3; do if [ $C = 2 ]; then BL=96; else BL=128; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c${C} -- python3 $R/bench.py --config $C --lanes 1 --batch $BL --no-cpu-baseline --no-h2d --steps 20 --warmup 2 > $R/gpurun_out/${TAG}_c${C}.json 2>/dev/null; done
