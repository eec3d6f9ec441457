"""Summarise the FETCH_SIZE / WRITE_SIZE rocprofv3 --pmc passes of tools/prof_rNN.sh into profiles-style files: HBM bytes per launch of the
cross-attention kernel (decode_attn_kernel<bf16, 8, RAGGED>), corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 128-B
requests of a wide coalesced stream at 64 B: x 2; WRITE_SIZE exact), separate passes.   python3 tools/pmc_summarise.py <dir> <tag>"""
import csv
import glob
import json
import os
import statistics
import sys

root, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
out = {}
rows_csv = ["kernel,counter,launches,median_KB,min_KB,max_KB"]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, f"pmc_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "decode_attn_kernel" in n and "true" in n and r["Counter_Name"] == ctr:   # RAGGED = true: the cross-attention form
                vals.append(float(r["Counter_Value"]))
    if vals:
        out[f"{ctr}_KB_median"], out[f"{ctr}_launches"] = statistics.median(vals), len(vals)
        rows_csv.append(f'"decode_attn_kernel<bf16,8,RAGGED>",{ctr},{len(vals)},{statistics.median(vals)},{min(vals)},{max(vals)}')
if "FETCH_SIZE_KB_median" in out and "WRITE_SIZE_KB_median" in out:
    out["hbm_bytes_per_launch"] = out["FETCH_SIZE_KB_median"] * 1024 * 2 + out["WRITE_SIZE_KB_median"] * 1024
    out["algorithmic_bytes_per_launch"] = 8 * 4096 * 2 * 1024 * 2
    out["kernel"] = "decode_attn_kernel<bf16, 8, RAGGED> (cross-attention K/V stream), batch 8 x S=4096, one layer per launch, 4 splits of 1024 keys"
    out["command"] = f"tools/prof_{tag}.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --legs ''"
    out["correction"] = "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE x 1024"
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, f"{tag}_pmc_cross_attn.json"), "w"), indent=1)
open(os.path.join(root, f"{tag}_pmc_traffic_summary.csv"), "w").write("\n".join(rows_csv) + "\n")
