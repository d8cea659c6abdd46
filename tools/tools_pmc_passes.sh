B="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcp1 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcp2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum --kernel-trace --output-format csv -d gpurun_out/pmc_ta -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY --kernel-trace --output-format csv -d gpurun_out/pmc_grbm -- $B > /dev/null 2>&1
ls gpurun_out/pmc_*/*/ | head -40
