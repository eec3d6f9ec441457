# Timing ablations of the one-pass attention backward (attn_bwd1p.hip, -DACAI_1P_ABL=bits; variants built by tools/build_variant.sh pN "... -DACAI_1P_ABL=N"
# attn_bwd1p.hip) and its SQ counters: bash tools/ab_bwd1p.sh  -> gpurun_out/ab_bwd1p/
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_bwd1p; mkdir -p $O
export ACAI_BENCH_ATTN_ONLY=mae-decoder
V=$R/acai_omr_amd/csrc/variants
for v in main ${ACAI_AB_VARIANTS:-p2 p4 p16 p32} main; do
  if [ $v = main ]; then L=$R/acai_omr_amd/csrc/libacai_omr_hip.so; else L=$V/$v.so; fi
  echo "== $v"; ACAI_OMR_LIB=$L timeout -k 10 100 python3 tools/bench_attn.py 10 2>&1 | grep prescaled || exit 1
done | tee $O/ablation.txt
ACAI_ATTN_BWD_1P=0 timeout -k 10 100 python3 tools/bench_attn.py 10 2>&1 | grep prescaled | sed 's/^/two kernels: /' | tee -a $O/ablation.txt
bash tools/pmc_attn64.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_attn64/summary.txt $O/pmc_summary.txt; grep -i "bwd1p" $O/pmc_summary.txt | cut -c1-600
