# same-box A/B of library builds / knobs on the fp16 vocoder alone (per-launch means of the wide fused kernels)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/speech_inpainting_amd
out=gpurun_out/respair_ab.txt; : > $out
run() {  # label, lib, extra env
  rm -rf /tmp/prof_ab
  env SI_HIP_LIB=$2 $3 true
  export SI_HIP_LIB=$2
  if [ -n "$3" ]; then export $3; fi
  rocprofv3 --kernel-trace -d /tmp/prof_ab -o t --output-format csv -- python3 tools/exp_vocoder_only.py 6 > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; exit 1; }
  if [ -n "$3" ]; then unset ${3%%=*}; fi
  echo "== $1" >> $out
  python3 tools/trace_respair.py $(find /tmp/prof_ab -name '*kernel_trace.csv') | grep -A9 "wide_kernel" | grep -v "^--" >> $out
}
for rep in 1 2; do
  run old $L/libsi_hip_old.so ""
  run new $L/libsi_hip.so ""
  run new_2wg $L/libsi_hip.so SI_RP_C128_2WG=1
done
cat $out
