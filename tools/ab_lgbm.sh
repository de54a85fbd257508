# per-launch lingemm durations (55 launches of a step, in order) with the tile height forced / chosen by the rule
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 64 96; do
  SI_LG_BM=$v timeout -k 10 300 python -m pytest tests/test_gpu_respair.py tests/test_gpu_parity.py -x -q -k "lingemm or bf16 or headline" > gpurun_out/lgbm_t_$v.log 2>&1 || { tail -30 gpurun_out/lgbm_t_$v.log; exit 1; }
  tail -1 gpurun_out/lgbm_t_$v.log
done
out=gpurun_out/lgbm_ab.txt; : > $out
for v in 128 96 64 0; do
  export SI_LG_BM=$v
  rm -rf /tmp/prof_l
  rocprofv3 --kernel-trace -d /tmp/prof_l -o t --output-format csv -- python3 bench.py --steps 6 --warmup 2 --cpu-clips 0 --no-fp32-leg > /tmp/b.log 2>/dev/null
  echo "SI_LG_BM=$v" >> $out
  python3 tools/trace_family.py $(find /tmp/prof_l -name '*kernel_trace.csv') lingemm 55 | sed -n 2p | cut -c1-200 >> $out
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.log | sed -n 1p >> $out
done
cat $out
