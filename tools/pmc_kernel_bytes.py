"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes):
    python3 tools/pmc_kernel_bytes.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <kernel-name regex> [grid-size filter]
Prints, per matching kernel name (template arguments shortened) and grid size: launches, median fetched bytes (FETCH_SIZE KB x 1024 x 2: gfx950
tallies the 128-B requests of a wide coalesced stream at 64 B), median written bytes (WRITE_SIZE KB x 1024), median duration under the
profiler and the bandwidth the two add up to."""
import collections
import csv
import glob
import os
import re
import statistics
import sys

root, pat = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, f"pmc_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not pat.search(n) or r["Counter_Name"] != ctr:
                continue
            short = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("unsigned short", "bf16").replace("void ", "")
            key = (short[:90], int(r["Grid_Size"]))
            acc[key][ctr].append(float(r["Counter_Value"]))
            acc[key]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(f"{'kernel':90s} {'grid':>10s} {'launches':>8s} {'fetch MB':>10s} {'write MB':>10s} {'us':>9s} {'TB/s':>6s}")
for (k, grid), d in sorted(acc.items(), key=lambda kv: -sum(kv[1]["ns"])):
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    fb, wb = statistics.median(d["FETCH_SIZE"]) * 1024 * 2, statistics.median(d["WRITE_SIZE"]) * 1024
    us = statistics.median(d["ns"]) / 1e3
    print(f"{k:90s} {grid:10d} {len(d['FETCH_SIZE']):8d} {fb / 1e6:10.1f} {wb / 1e6:10.1f} {us:9.1f} {(fb + wb) / us / 1e6:6.2f}")
