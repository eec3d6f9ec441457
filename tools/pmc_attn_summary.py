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
            if "attn_" not in k and "bwd1p" not in k:
                continue
            m = re.search(r"(attn_\w+)<([^>]*)>", k)
            short = m.group(1) + "<" + m.group(2).replace("unsigned short", "bf16") + ">" if m else re.sub(r"\(anonymous namespace\)::", "", k).split("(")[0][:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[short]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    print("==", os.path.basename(d))
    for k, cs in acc.items():
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        extra = {}
        if "GRBM_GUI_ACTIVE" in mean:   # effective clock (MI355X_MICROARCH.md, DVFS give-back): the counter is summed over the 8 XCDs
            extra["clock_GHz"] = f"{mean['GRBM_GUI_ACTIVE'] / 8 / (mean['dur_us'] * 1e3):.3f}"
        print("  ", k, {c.replace("SQ_", ""): f"{v:.4g}" for c, v in sorted(mean.items())}, extra, "n=", len(next(iter(cs.values()))))
