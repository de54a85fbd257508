# same-box A/B: whole-resblock kernel on the C = 32 stage (SI_VOC_CHAIN=1) vs one launch per pair (=0); bench per-family tables
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_respair.py -x -q -s -k "chain or fused" > gpurun_out/chain_t.log 2>&1 || { tail -30 gpurun_out/chain_t.log; exit 1; }
grep "B=\|saturated\|passed" gpurun_out/chain_t.log
out=gpurun_out/chain_ab.txt; : > $out
for rep in 1 2; do for v in 0 1; do
  echo "== SI_VOC_CHAIN=$v" >> $out
  SI_VOC_CHAIN=$v python3 bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-fp32-leg > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json | sed -n 1p >> $out
  grep -i "respair_f16_c32\|reschain\|respair_f16_c64" /tmp/b.err >> $out
done; done
cat $out
SI_HIP_LIB=$GRAFT_REPO_ROOT/speech_inpainting_amd/libsi_hip_timeline.so timeout -k 10 200 python tools/exp_vocoder_only.py 6 2>&1 | grep chain
