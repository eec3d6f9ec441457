# rocprofv3 kernel stats of the MAE and teacher-forced training legs: bash tools/prof_train.sh [tag]  -> gpurun_out/prof_train_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_train_${1:-x}
mkdir -p $O
for L in ${ACAI_PROF_LEGS:-mae tf}; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$L -o $L -- python3 $R/tools/prof_leg.py $L > $O/$L.log 2>&1 || echo $L failed
  tail -1 $O/$L.log | cut -c1-200
done
rm -f $(find $O -name "*kernel_trace.csv")
for L in ${ACAI_PROF_LEGS:-mae tf}; do f=$(find $O/$L -name "*kernel_stats.csv" | head -1); echo "== $L"; python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n = r["Name"]
    m = re.search(r"(\w+)<([^>]*)>", n)
    short = (m.group(1) + "<" + m.group(2)[:40] + ">") if m else n[:60]
    print(f'{short:70s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total", tot / 1e6, "ms")
PY
done
