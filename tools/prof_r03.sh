# Round-3 profiles (on the GPU box): bash tools/prof_r03.sh  -> gpurun_out/prof_r03/ ; summaries are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode -o decode -- python3 $R/bench.py --no-cpu-baseline --legs "" > $O/decode.log 2>&1 || echo decode failed
for L in mae tf ragged; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$L -o $L -- python3 $R/tools/prof_leg.py $L > $O/$L.log 2>&1 || echo $L failed
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o r --output-format csv -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --legs "" > $O/log_$C.txt 2>&1 || echo "$C failed"
done
python3 $R/tools/pmc_summarise.py $O r03 > $O/pmc_summary.log 2>&1
for L in decode mae tf ragged; do tail -1 $O/$L.log | cut -c1-300; done
find $O -name "*kernel_stats.csv" | head
rm -f $(find $O -name "*kernel_trace.csv") $(find $O -name "*counter_collection.csv")   # keep the merged scratch small
