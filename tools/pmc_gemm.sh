cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $O/$T -o r --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/$T.log 2>&1 || echo "$T failed"
done
python3 - $O <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tools")
names = ["dec qkv fwd 131072x1536x512", "dec lin1 gelu 131072x3072x512", "dec lin2 131072x512x3072", "enc qkv 32768x2304x768", "dec da gelu' 131072x3072x512",
         "dec out +res 131072x512x512"]
for d in sorted(glob.glob(root + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_nt" in r["Kernel_Name"]]
        by = collections.defaultdict(list)
        for r in rows:
            by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"][:60]))
        for c, v in by.items():
            v.sort()
            print(c, [f"{names[i // 4] if i % 4 == 0 else ''} {x[1]:.4g}" for i, x in enumerate(v)])
PY
rm -f $(find $O -name "*kernel_trace.csv")
