# A/B: row padding of the bf16 FFN intermediate (SI_ENC_FFNPAD elements) -- L2 channel spread of FFN2's A rows
cd "$(dirname "$0")/.."
for f in 0 64 0 64 32 128; do
  echo "=== SI_ENC_FFNPAD=$f"
  SI_ENC_FFNPAD=$f timeout -k 10 200 python tools/exp_encoder_only.py 10 | grep -E "encoder alone|x3072 |768x3072|checksum"
done
