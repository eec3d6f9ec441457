# Round-4 profiles (on the GPU box): bash tools/prof_r04.sh [tag]  -> gpurun_out/prof_r04<tag>/ ; summaries are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04$1
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode -o decode -- python3 $R/bench.py --no-cpu-baseline --legs "" > $O/decode.log 2>&1 || echo decode failed
for L in mae tf ragged; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$L -o $L -- python3 $R/tools/prof_leg.py $L > $O/$L.log 2>&1 || echo $L failed
done
export ACAI_BENCH_NO_GRAPH=1   # counter collection over the 8-step decode graph segfaults inside rocprofv3: the PMC passes enqueue the same launches one by one
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o r --output-format csv -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --legs "" > $O/log_$C.txt 2>&1 || echo "$C failed"
done
unset ACAI_BENCH_NO_GRAPH
python3 $R/tools/pmc_summarise.py $O r04 > $O/pmc_summary.log 2>&1
# LayerNorm kernels of the MAE step: fetched / written bytes against the minimum (verdict r3 item 6)
mkdir -p $O/ln
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/ln/pmc_$C -o r --output-format csv -- python3 $R/tools/prof_leg.py mae > $O/ln/log_$C.txt 2>&1 || echo "ln $C failed"
done
python3 $R/tools/pmc_kernel_bytes.py $O/ln "layernorm_kernel|ln_bwd_fused" > $O/r04_ln_traffic.txt 2>&1
for L in decode mae tf ragged; do tail -1 $O/$L.log | cut -c1-300; done
cat $O/r04_ln_traffic.txt
find $O -name "*kernel_stats.csv" | head
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*counter_collection.csv")   # keep the merged scratch small
