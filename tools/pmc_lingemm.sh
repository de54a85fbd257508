cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() { tag=$1; shift; rm -rf /tmp/p_$tag; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p_$tag -- python3 $R/tools/exp_encoder_only.py 2 > /tmp/p_$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 /tmp/p_$tag.log; return; }; python3 $R/tools/pmc_by_grid.py /tmp/p_$tag/*/*_counter_collection.csv /tmp/p_$tag/*/*_kernel_trace.csv lingemm2 | head -3 > $R/gpurun_out/r3_lg_pmc_$tag.txt; }
run B SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run C SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES
run D TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run E TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
run F TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
cat $R/gpurun_out/r3_lg_pmc_*.txt
