# PMC passes over the decode step (cross-attention kernel is filtered by the summariser): bash tools/pmc_cross.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_cross
mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $O/p${i} -o r --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-mae --steps 32 --warmup 8 > $O/log_${i}.txt 2>&1 || echo "pass $i failed"
done
ls $O
