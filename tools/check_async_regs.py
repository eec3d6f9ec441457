"""Static check of the untracked asm loads in gemm_nt_pp_kernel (bias / deferred chunks): between a `global_load_dword[x4] vN, ...` issued from
inline asm and the next explicit take (`s_waitcnt vmcnt(4|0)` emitted by the kernel's own asm) no instruction may read or write the
destination registers - the compiler must not copy a register whose data has not landed.  Usage: python tools/check_async_regs.py gemm.s"""
import re
import sys

src = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r"^(_ZN12_GLOBAL__N_117gemm_nt_pp_kernel\w+):[^\n]*\n(.*?)s_endpgm", src, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    n_loads = 0
    for i, ln in enumerate(body):
        mm = re.match(r"\s*global_load_dword(x4)?\s+(v\[(\d+):(\d+)\]|v(\d+)),", ln)
        if not mm:
            continue
        n_loads += 1
        regs = set(range(int(mm.group(3)), int(mm.group(4)) + 1)) if mm.group(3) else {int(mm.group(5))}
        # scan forward to the next s_waitcnt vmcnt in layout order, through at most 400 lines
        for j in range(i + 1, min(i + 2000, len(body))):
            t = body[j]
            if re.search(r"s_waitcnt vmcnt\((0|4)\)", t):
                break
            if t.strip().startswith(";") or t.strip().startswith(".") or "global_load_dword" in t and j == i:
                continue
            used = set()
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                used |= set(range(int(a), int(b) + 1))
            for a in re.findall(r"\bv(\d+)\b", t):
                used.add(int(a))
            if used & regs:
                print(f"{name}: line {i}: {ln.strip()}  -> touched before the wait at +{j - i}: {t.strip()}")
                bad += 1
                break
    print(name, "untracked loads checked:", n_loads)
sys.exit(1 if bad else 0)
