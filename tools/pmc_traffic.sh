# HBM traffic of the cross-attention kernel from the PMC counters (separate passes, as MI355X_MICROARCH.md prescribes): bash tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_traffic
mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/$C -o r --output-format csv -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-mae > $O/log_$C.txt 2>&1 || echo "$C failed"
done
ls $O
