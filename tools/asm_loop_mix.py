"""Instruction mix of the largest loops of a kernel in a hipcc -S listing: python3 tools/asm_loop_mix.py file.s kernel-substring
Finds backward branches (s_cbranch_* to an earlier label), prints per loop: instruction counts by class."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
kern = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (not the first s_endpgm: early exits have their own)
labels = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m:
        labels[m.group(1)] = i
loops = []
for i in range(start, end):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i]) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_exp"): return "v_exp"
    if op.startswith("v_cvt_pk"): return "cvt_pk"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_"): return op.split("_")[0] + "_mem"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "salu"
    return "other"
for a, b in sorted(loops, key=lambda ab: ab[0] - ab[1])[:4]:
    c = collections.Counter(); others = collections.Counter()
    for l in lines[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].endswith(":") or t[0].startswith(".") or t[0].startswith(";"): continue
        k = cls(t[0]); c[k] += 1
        if k == "valu_other": others[t[0]] += 1
    print(f"loop lines {a}-{b} ({b - a} lines):", dict(c))
    print("   valu_other:", dict(others.most_common(12)))
