# same-box A/B: wide fused kernels on 128 x 64 wave tiles, one wave per SIMD (SI_RP_TALL bit 0: C = 128, bit 1: C = 256)
cd $GRAFT_REPO_ROOT
SI_RP_TALL=3 timeout -k 10 300 python -m pytest tests/test_gpu_respair.py -x -q -k "chain or fused" > gpurun_out/tall_t.log 2>&1 || { tail -20 gpurun_out/tall_t.log; exit 1; }
tail -1 gpurun_out/tall_t.log
for rep in 1 2; do for v in 0 1 2 3; do
  echo "== SI_RP_TALL=$v"
  SI_RP_TALL=$v python3 bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-fp32-leg > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json | sed -n 1p
  grep -i "respair_f16_c128\|respair_f16_c256" /tmp/b.err | sed -n 1,4p
done; done
