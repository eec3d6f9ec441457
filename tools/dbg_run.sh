#!/bin/bash
# Debug aid: the d_h = 32 backward through the one-pass kernel against the two-kernel form on the same inputs (tools/dbg_bwd64w.py), then the
# attention micro-benchmark both ways.   bash tools/dbg_run.sh [out_dir]
out=${1:-gpurun_out/dbg}
mkdir -p $out
for shape in "512 512 1" "1024 1024 2" "1000 1024 1" "4096 4096 16"; do
  ACAI_ATTN_BWD_1P=0 timeout -k 10 120 python tools/dbg_bwd64w.py run $out/a.pt $shape 32 &&
  ACAI_ATTN_BWD_1P=1 timeout -k 10 120 python tools/dbg_bwd64w.py run $out/b.pt $shape 32 &&
  echo "== $shape" && timeout -k 10 60 python tools/dbg_bwd64w.py cmp $out/a.pt $out/b.pt | grep -v "^ \|tensor\|\[" || exit 1
done
