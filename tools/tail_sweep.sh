# A/B of the launch-tail pieces (KANVIT_TAIL: 0 = off, -1 = automatic, k = k tail tiles) and of ChebyKAN's 16-row weight gradient
# (KANVIT_BW_NO_T16=1 = the 32-row form) on one MSA block at ViT-B; all arms in ONE process sequence on one box
set -e
for k in 0 -1 10 14; do
  echo "== KANVIT_TAIL=$k"
  KANVIT_TAIL=$k python tools/time_op.py cheby b=128 2>/dev/null | grep "qkv_"
done
echo "== KANVIT_BW_NO_T16=1"
KANVIT_BW_NO_T16=1 python tools/time_op.py cheby b=128 2>/dev/null | grep "qkv_bwd_weight"
