# Same-box A/B of the whole configs[1] step: this round's second-half changes off (the environment knobs that select the former
# kernels) against the default build.  usage: bash tools/ab_round4.sh
cd "$(dirname "$0")/.."
OLD="SI_ENC_GEMMCU=0 SI_ENC_LNFUSE=0 SI_VOC_UPSGEMM=0 SI_ENC_FFNPAD=0"
for rep in 1 2 3; do
  for mode in old new; do
    if [ $mode = old ]; then pre="env $OLD"; else pre=""; fi
    $pre python bench.py --steps 10 --warmup 3 --cpu-clips 0 --no-config-legs --no-fp32-leg --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['ms_per_step'], d['value'])"
  done
done
