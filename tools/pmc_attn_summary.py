"""Per-kernel means of the SQ counters collected by tools/ab_attn.sh: python tools/pmc_attn_summary.py <dir>"""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pmc*_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn_" not in k:
                continue
            m = re.search(r"(attn_\w+)<([^>]*)>", k)
            short = m.group(1) + "<" + m.group(2).replace("unsigned short", "bf16") + ">" if m else k[:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", os.path.basename(d))
    for k, cs in acc.items():
        print("  ", k, {c.replace("SQ_", ""): f"{sum(v) / len(v):.4g}" for c, v in sorted(cs.items())}, "n=", len(next(iter(cs.values()))))
