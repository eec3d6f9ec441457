"""GPU busy fraction from a rocprofv3 --kernel-trace csv: python tools/gpu_idle_from_trace.py <kernel_trace.csv> [marker kernel substring]
Splits the trace at every launch of the marker kernel (default: adamw_kernel = one per MAE step) and prints, per step, the wall span, the
sum of kernel durations and the idle time between kernels (gaps > 20 us listed)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
cuts = [i for i, e in enumerate(ev) if marker in e[2]]
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = ev[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e[1] - e[0] for e in seg)
    gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)]
    big = sorted([g for g in gaps if g[0] > 20000], reverse=True)[:8]
    print(f"step: span {span/1e6:.2f} ms, kernels {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms over {len(seg)} launches; gaps > 20 us: {len([g for g in gaps if g[0] > 20000])}")
    for g in big:
        print(f"    {g[0]/1e3:8.1f} us between {g[1]} -> {g[2]}")
