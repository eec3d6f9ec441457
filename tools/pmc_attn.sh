# PMC passes over the varlen attention kernels (MAE-decoder shape): bash tools/pmc_attn.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn
mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  ACAI_BENCH_ATTN_ONLY=mae-decoder timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C -d $O/p${i} -o r --output-format csv -- python3 $R/tools/bench_attn.py 1 > $O/log_${i}.txt 2>&1 || echo "pass $i failed"
done
ls $O
