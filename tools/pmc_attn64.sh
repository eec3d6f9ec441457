# SQ counters of the d_h = 64 attention kernels (teacher-forced encoder shape): bash tools/pmc_attn64.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn64
mkdir -p $O
export ACAI_BENCH_ATTN_ONLY=${ACAI_BENCH_ATTN_ONLY:-tf-encoder}
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pmcA_main -o r --output-format csv -- python3 $R/tools/bench_attn.py 2 > $O/pmcA.log 2>&1 || echo "pmcA failed"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES -d $O/pmcB_main -o r --output-format csv -- python3 $R/tools/bench_attn.py 2 > $O/pmcB.log 2>&1 || echo "pmcB failed"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmcC_main -o r --output-format csv -- python3 $R/tools/bench_attn.py 2 > $O/pmcC.log 2>&1 || echo "pmcC failed"
python3 $R/tools/pmc_attn_summary.py $O > $O/summary.txt 2>&1; cut -c1-600 $O/summary.txt
rm -f $(find $O -name "*kernel_trace.csv")
