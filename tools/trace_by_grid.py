"""Per (kernel, grid size) launch statistics from a rocprofv3 --kernel-trace CSV (one tool run over several shapes keeps them apart):
    python3 tools/trace_by_grid.py <dir or *_kernel_trace.csv> [name regex]"""
import collections, csv, glob, os, re, statistics, sys
src = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if pat and not pat.search(n):
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("unsigned short", "bf16").replace("void ", "")
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])) if "Grid_Size_X" in r else (int(r["Grid_Size"]),)
        acc[(short[:80], g)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(f"{'kernel':80s} {'grid (workgroups)':>22s} {'n':>5s} {'median us':>10s} {'min us':>9s}")
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:80s} {str(g):>22s} {len(v):5d} {statistics.median(v) / 1e3:10.1f} {min(v) / 1e3:9.1f}")
