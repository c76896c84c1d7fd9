# A/B of diagnostic builds of the 16-row-tile attention backward (csrc/attention16.hip, -DB16_ABLATE=n drops groups of MFMAs: what is left
# is the protocol, the LDS traffic and the fills) -- one process sequence on one box.  Build first: for n in 7 16 32 48 55; do tools/build_variant.sh abl$n -DB16_ABLATE=$n; done
for n in "" 48 55; do
  L=""; [ -n "$n" ] && L=kan-vit_amd/kanvit/_ab/libkanvit_abl$n.so
  echo "== ablate=${n:-0}"
  KANVIT_LIB=$L python tools/time_op.py cheby 2>/dev/null | grep "attn_bwd"
done
