cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  for S in "4096 4096 4096" "131072 1536 512"; do
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C -d $O/p${i}_$(echo $S | tr ' ' 'x') -o r --output-format csv -- python3 $R/tools/prof_gemm.py $S 3 > $O/log_${i}.txt 2>&1 || echo "pass $i $S failed"
  done
done
ls $O
