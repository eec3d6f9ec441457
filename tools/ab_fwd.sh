R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
V=$R/acai_omr_amd/csrc/variants
export ACAI_BENCH_ATTN_ONLY=mae-decoder
for rep in 1 2; do
for v in ${ACAI_AB_VARIANTS:-main f_w3 f_w4}; do
  if [ $v = main ]; then L=$R/acai_omr_amd/csrc/libacai_omr_hip.so; else L=$V/$v.so; fi
  echo "== $v"; ACAI_OMR_LIB=$L timeout -k 10 200 python3 tools/bench_attn.py 10 2>&1 | grep "prescaled" | grep -v fp32
done; done
