cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_prof
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode -o decode -- python3 $R/bench.py --no-cpu-baseline --no-mae > $O/decode.log 2>&1 || echo decode failed
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mae -o mae -- python3 $R/tools/prof_mae.py 32 3 > $O/mae.log 2>&1 || echo mae failed
ls $O/decode $O/mae
tail -1 $O/decode.log | cut -c1-200
tail -1 $O/mae.log | cut -c1-200
