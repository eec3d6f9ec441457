cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/cross -o cross -- python3 $R/tools/bench_cross_train_attn.py > $O/cross.log 2>&1 || echo cross failed
python3 $R/tools/trace_by_grid.py $O/cross "attn" > $O/cross_by_grid.txt 2>&1; cat $O/cross_by_grid.txt
timeout -k 10 200 python3 $R/tools/bench_mae_gemms.py > $O/mae_gemms.txt 2>&1; cat $O/mae_gemms.txt
rm -f $(find $O -name "*kernel_trace.csv")
