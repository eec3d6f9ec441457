# A/B of attention kernel variants on the GPU box: bash tools/ab_attn.sh  (variants built by tools/build_variant.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_attn
mkdir -p $O
V=$R/acai_omr_amd/csrc/variants
libof() { if [ "$1" = main ]; then echo $R/acai_omr_amd/csrc/libacai_omr_hip.so; else echo $V/$1.so; fi; }
cd $R
for v in ${ACAI_AB_TEST_VARIANTS:-main}; do
  ACAI_OMR_LIB=$(libof $v) timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -m gpu -q -k "attn or mae_forward" > $O/test_$v.log 2>&1 || { echo "TEST FAILED $v"; grep -a "^FAILED\|^ERROR\|Error" $O/test_$v.log | cut -c1-300 | tail -30; }
  tail -1 $O/test_$v.log
done
for v in ${ACAI_AB_VARIANTS:-main s3 s2}; do
  echo "== $v"
  ACAI_OMR_LIB=$(libof $v) timeout -k 10 200 python3 tools/bench_attn.py 10 2>&1 | tee $O/bench_$v.log | grep -v Warn
done
cd /tmp
export ACAI_BENCH_ATTN_ONLY=${ACAI_BENCH_ATTN_ONLY:-mae-decoder}
for v in ${ACAI_AB_PMC_VARIANTS:-main}; do
  export ACAI_OMR_LIB=$(libof $v)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pmcA_$v -o r --output-format csv -- python3 $R/tools/bench_attn.py 2 > $O/pmcA_$v.log 2>&1 || echo "pmcA $v failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES -d $O/pmcB_$v -o r --output-format csv -- python3 $R/tools/bench_attn.py 2 > $O/pmcB_$v.log 2>&1 || echo "pmcB $v failed"
done
python3 $R/tools/pmc_attn_summary.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
rm -f $(find $O -name "*kernel_trace.csv")
cd $R && timeout -k 10 300 python3 bench.py --legs mae,tf --no-cpu-baseline > $O/bench_legs.log 2>&1; tail -1 $O/bench_legs.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('mae', d.get('mae'), 'tf', d.get('tf_step'))" || tail -5 $O/bench_legs.log
