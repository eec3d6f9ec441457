# bash tools/dbg_run.sh : the wide d_h = 64 backward forms (ACAI_ATTN64_BWD_WIDE = 1 dQ, 2 dK/dV, 3 both) against the one-block kernels (0), per
# 32 x 32 block of each gradient (tools/dbg_bwd64w.py).  This is the tool that showed the VALU-write -> asm-MFMA hazard of round 4 (a few blocks wrong
# without the s_nop in front of the inline-asm MFMAs, exact with it).
O=gpurun_out/dbg; mkdir -p $O
for SH in "256 448 1" "600 1000 2"; do
  set -- $SH
  ACAI_ATTN64_BWD_WIDE=0 python tools/dbg_bwd64w.py run $O/ref.pt $1 $2 $3 64 || exit 1
  for W in 1 2 3; do
    ACAI_ATTN64_BWD_WIDE=$W python tools/dbg_bwd64w.py run $O/w$W.pt $1 $2 $3 64 || exit 1
    echo "== shape $SH  wide form $W"; python tools/dbg_bwd64w.py cmp $O/ref.pt $O/w$W.pt | grep "max diff"
  done
done
