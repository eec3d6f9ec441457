cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_tf
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tf -o tf -- python3 $R/tools/bench_tf_and_ragged.py tf > $O/tf.log 2>&1 || echo tf failed
grep '^{' $O/tf.log | cut -c1-400
head -16 $O/tf/tf_kernel_stats.csv | cut -c1-150
