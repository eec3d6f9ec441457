# bash tools/dbg_run.sh : the one-pass d_h = 32 backward against the two-pass kernels
O=gpurun_out/d5; mkdir -p $O
for SH in "512 512 1" "1024 1536 2" "576 700 1"; do
  set -- $SH
  ACAI_ATTN_BWD_1P=0 python tools/dbg_bwd64w.py run $O/ref.pt $1 $2 $3 32 || exit 1
  ACAI_ATTN_BWD_1P=1 python tools/dbg_bwd64w.py run $O/w.pt $1 $2 $3 32 || exit 1
  echo "== shape $SH"; python tools/dbg_bwd64w.py cmp $O/ref.pt $O/w.pt | grep "max diff"
done
python tools/dbg_bwd64w.py cmp $O/ref.pt $O/w.pt | head -60
