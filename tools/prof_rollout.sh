cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_roll
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -o r -- python3 $R/tools/bench_rollout.py > $O/r.log 2>&1 || echo failed
grep -v "^W\|^E\|amdgpu.ids" $O/r.log | tail -4 | cut -c1-300
head -12 $O/r/r_kernel_stats.csv | cut -c1-160
