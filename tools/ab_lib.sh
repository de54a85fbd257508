cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/speech_inpainting_amd
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/ab_t.log 2>&1 || { tail -20 gpurun_out/ab_t.log; exit 1; }
tail -1 gpurun_out/ab_t.log
for rep in 1 2 3; do for v in libsi_hip_A.so libsi_hip.so; do
  echo "== $v"
  SI_HIP_LIB=$L/$v python3 bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-fp32-leg > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json | sed -n 1p
  grep -i "attention" /tmp/b.err | sed -n 1,2p
done; done
