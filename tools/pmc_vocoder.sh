# rocprofv3 counter passes over the fp16 vocoder alone (tools/exp_vocoder_only.py, B = 32), per kernel: wave-time shares, matrix-pipe
# busy, LDS / vector-memory issue stalls, L1 requests.  Separate --pmc passes, --kernel-trace only.  Run on the GPU box from the repo
# root; writes gpurun_out/r03_vocoder_pmc.txt.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r03_vocoder_pmc.txt; : > $out
run() { tag=$1; shift; rm -rf /tmp/p_$tag; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p_$tag -- python3 $R/tools/exp_vocoder_only.py 2 > /tmp/p_$tag.log 2>&1 || { echo "pass $tag failed" >> $out; tail -3 /tmp/p_$tag.log >> $out; return; }; echo "== $tag: $*" >> $out; python3 $R/tools/pmc_by_grid.py /tmp/p_$tag/*/*_counter_collection.csv /tmp/p_$tag/*/*_kernel_trace.csv res >> $out; }
run wave SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run issue SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run lds SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
wc -l $out
