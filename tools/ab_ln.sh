cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/ln_t.log 2>&1 || { tail -30 gpurun_out/ln_t.log; exit 1; }
tail -2 gpurun_out/ln_t.log
out=gpurun_out/ln_ab.txt; : > $out
for rep in 1 2; do for v in 100000 8 16 32; do
  echo "== SI_LN_WAVES=$v" >> $out
  SI_LN_WAVES=$v python3 bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-fp32-leg > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json | sed -n 1p >> $out
  grep -i "layernorm" /tmp/b.err | sed -n 1p >> $out
done; done
cat $out
