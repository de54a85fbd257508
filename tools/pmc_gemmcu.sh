# rocprofv3 counter passes over the bf16 encoder alone (tools/exp_encoder_only.py, B = 32 x 4 s) for gemmcu.hip's launches, per kernel
# instantiation and grid size (= GEMM shape).  Separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md).  Run on the GPU box
# from the repo root; writes gpurun_out/r04_gemmcu_pmc.txt.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r04_gemmcu_pmc.txt; : > $out
run() { tag=$1; shift 1; rm -rf /tmp/p_$tag; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p_$tag -- python3 $R/tools/exp_encoder_only.py 2 > /tmp/p_$tag.log 2>&1 || { echo "pass $tag failed" >> $out; tail -3 /tmp/p_$tag.log >> $out; return; }; echo "== $tag: $*" >> $out; python3 $R/tools/pmc_by_grid.py /tmp/p_$tag/*/*_counter_collection.csv /tmp/p_$tag/*/*_kernel_trace.csv "gemmcu_kernel" >> $out 2>&1; }
run wave SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run issue SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
wc -l $out
