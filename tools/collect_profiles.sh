#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run from the repo root):  tools/collect_profiles.sh r02
# 1. --kernel-trace --stats of the default bench (per-kernel time; the bench's own JSON line and per-family table)
# 2. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM traffic), GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES (clock, pipe)
# The program itself follows `--` (python3 ...), never a wrapper.  Outputs land in gpurun_out/<tag>_*; copy what should
# be judged into profiles/.
set -o pipefail
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps /tmp/pf /tmp/pw /tmp/pc
# (--no-config-legs: the configs[3] / configs[4] legs launch the same kernel instantiations on other shapes and would dilute the
#  per-kernel averages that roofline.avg_launch_ms is checked against; they get their own pass below)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python3 $root/bench.py --steps 5 --warmup 2 --cpu-clips 0 --no-config-legs > $out/${tag}_bench.json 2> $out/${tag}_bench_stderr.txt || exit 1
cp /tmp/ps/*/*_kernel_stats.csv $out/${tag}_kernel_stats.csv
cp /tmp/ps/*/*_agent_info.csv $out/${tag}_agent_info.csv
rm -rf /tmp/pl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python3 $root/bench.py --only-config-legs --steps 6 > $out/${tag}_legs.json 2> $out/${tag}_legs_stderr.txt || exit 1
cp /tmp/pl/*/*_kernel_stats.csv $out/${tag}_legs_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $root/bench.py --steps 1 --warmup 1 --cpu-clips 0 --no-kernel-events --no-fp32-leg --no-config-legs > /tmp/pf.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $root/bench.py --steps 1 --warmup 1 --cpu-clips 0 --no-kernel-events --no-fp32-leg --no-config-legs > /tmp/pw.log 2>&1 || exit 1
cd $root/tools && python3 collect_traffic.py /tmp/pf/*/*_counter_collection.csv /tmp/pw/*/*_counter_collection.csv $out/${tag}_hbm_traffic.json
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pc -- python3 $root/bench.py --steps 2 --warmup 1 --cpu-clips 0 --no-kernel-events --no-fp32-leg --no-config-legs > /tmp/pc.log 2>&1 || exit 1
cd $root/tools && python3 collect_pmc.py /tmp/pc/*/*_counter_collection.csv /tmp/pc/*/*_kernel_trace.csv $out/${tag}_clock_mfma.json > $out/${tag}_clock_mfma.txt
echo "profiles collected: $out/${tag}_*"
