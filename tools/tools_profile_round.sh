# usage (repo root, on the GPU box): bash tools/tools_profile_round.sh <tag>   -> gpurun_out/<tag>_*
T=$1
B="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline"   # two lanes, 128 frames per launch
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_pmcF -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${T}_pmcW -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d gpurun_out/${T}_pmcT -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/${T}_pmcS -- $B > /dev/null 2>&1
ls gpurun_out/ | grep ${T}
tail -c 600 gpurun_out/${T}_bench.json
