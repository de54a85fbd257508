# A/B of gemmcu.hip's instantiations on the bench's encoder (SI_ENC_GEMMCU = 10 + c forces instantiation c); usage: bash tools/ab_gemmcu.sh "1 14 15" [timeline]
set -e
cd "$(dirname "$0")/.."
for f in ${1:-0 1 10 11 12 13}; do
  echo "=== SI_ENC_GEMMCU=$f $2"
  if [ "$2" = "timeline" ]; then export SI_HIP_LIB=$PWD/speech_inpainting_amd/libsi_hip_timeline.so; fi
  SI_ENC_GEMMCU=$f timeout -k 10 200 python tools/exp_encoder_only.py 10 | grep -E "encoder alone|gemm|checksum|timeline"
done
