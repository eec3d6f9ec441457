# bash tools/dbg_run.sh : the wide d_h = 64 backward of each variant library against the one-block kernels
O=gpurun_out/d3; mkdir -p $O
ACAI_ATTN64_BWD_WIDE=0 python tools/dbg_bwd64w.py run $O/ref.pt 256 448 1 || exit 1
for V in bw_pre7 bw_post7 bw_pre3 bw_post3 bw_pre1 bw_post1; do
  if [ -n "$V" ]; then export ACAI_OMR_LIB=$PWD/acai_omr_amd/csrc/variants/$V.so; fi
  ACAI_ATTN64_BWD_WIDE=3 python tools/dbg_bwd64w.py run $O/w_$V.pt 256 448 1 || exit 1
  echo "== variant '$V'"; python tools/dbg_bwd64w.py cmp $O/ref.pt $O/w_$V.pt | grep "max diff"
done
