"""Static checks of the generated gfx950 assembly for the two places where a kernel's correctness rests on something the compiler does
not model.  `_lib.build()` runs them on every (re)compile of the sources concerned and FAILS THE BUILD on a hit; `tools/check_async_regs.py`
is the command-line form.

1. gemm_nt_pp_kernel / gemm_tn_pp_kernel (gemm.hip): vector-memory loads issued from inline asm (`global_load_dword[x4] vN, ...`: the bias of
   the tile in flight, the deferred GELU chunks) are invisible to the wait-count pass; their destination registers hold garbage until the
   kernel's own counted `s_waitcnt vmcnt(N)` retires them.  Between such a load and the next `s_waitcnt ... vmcnt(` in layout order no
   instruction may read or write those registers - a register-allocator copy, spill or reuse there reads data that has not landed (the
   miscompile an earlier version of the kernel hit).
2. attn_fwd64w_kernel (attn_fwd64w.hip): the tile barrier waits `lgkmcnt(4)` instead of draining the LDS queue - it relies on EXACTLY the four
   fragment reloads of the last two MFMA gaps being issued behind the staged tiles' ds_write_b128.  More LDS operations there and the counted
   wait no longer covers the stores.
3. attn_fwd64w.hip / attn_bwd64w.hip: MFMAs issued from inline asm (accumulator-register C / D).  A VALU write of a register needs two wait
   states before an MFMA reads it; hipcc pads its own MFMAs and knows nothing about an asm one.  Round 4 hit this for real: the compiler's
   v_accvgpr_mov / v_accvgpr_write copies that assemble an operand tuple sat directly in front of the asm statement and a few 32 x 32 blocks of
   the d_h = 64 gradients came out wrong (tools/dbg_bwd64w.py).  Every asm MFMA: no v_* instruction may write one of its source registers in
   the two wait states in front of it (an `s_nop N` counts N + 1, any other instruction 1).
"""
import re

_VREG = re.compile(r"\bv(\d+)\b")
_VRANGE = re.compile(r"v\[(\d+):(\d+)\]")


def _regs_of(text):
    used = set()
    for a, b in _VRANGE.findall(text):
        used |= set(range(int(a), int(b) + 1))
    for a in _VREG.findall(text):
        used.add(int(a))
    return used


def _functions(src, pattern):
    # (up to the function's end label, not its first s_endpgm: an early return has its own)
    for m in re.finditer(r"^(" + pattern + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:", src, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def check_untracked_loads(src, kernels=r"_ZN12_GLOBAL__N_1\d+gemm_(?:nt|tn)_pp_kernel"):
    """Returns (problems, n_loads): problems = list of strings, one per asm load whose destination is touched before the next vmcnt wait."""
    problems, n_loads = [], 0
    for name, body in _functions(src, kernels):
        in_asm = False
        for i, ln in enumerate(body):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            mm = re.match(r"global_load_dword(?:x[234])?\s+(v\[\d+:\d+\]|v\d+),", t)
            if not (in_asm and mm):
                continue   # (loads the compiler issued itself are tracked by its own wait counts)
            n_loads += 1
            regs = _regs_of(mm.group(1))
            waited = False
            for j in range(i + 1, len(body)):
                u = body[j].strip()
                if not u or u.startswith(";") or u.startswith("."):
                    continue
                if re.search(r"s_waitcnt\b.*vmcnt\(\d+\)", u):
                    waited = True
                    break
                if _regs_of(u) & regs:
                    problems.append(f"{name}: +{i}: `{t}` -> destination touched {j - i} lines later, before any vmcnt wait: `{u}`")
                    waited = True
                    break
            if not waited:
                problems.append(f"{name}: +{i}: `{t}` -> no vmcnt wait follows it in the function")
    return problems, n_loads


def check_fwd64w_barrier(src, kernels=r"_ZN12_GLOBAL__N_118attn_fwd64w_kernel", allowed=4):
    """Every `s_waitcnt lgkmcnt(4)` + `s_barrier` pair: at most `allowed` LDS instructions between the last ds_write before it and the wait."""
    problems, n = [], 0
    for name, body in _functions(src, kernels):
        for i, ln in enumerate(body):
            if "s_barrier" not in ln:
                continue
            k = i - 1
            while k >= 0 and (not body[k].strip() or body[k].strip().startswith(";") or re.match(r"\s*s_(?!waitcnt)", body[k])):
                k -= 1
            if k < 0 or not re.search(r"s_waitcnt lgkmcnt\(%d\)\s*$" % allowed, body[k].strip()):
                continue   # a barrier behind a full drain (prologue, slow path): nothing to count
            n += 1
            lds_after = 0
            for j in range(k - 1, -1, -1):
                u = body[j].strip()
                if u.startswith("ds_write") or u.startswith("ds_store"):
                    break
                if u.startswith("ds_"):
                    lds_after += 1
                if "s_barrier" in u:
                    problems.append(f"{name}: +{i}: no ds_write between this counted barrier and the previous barrier")
                    break
            if lds_after > allowed:
                problems.append(f"{name}: +{i}: {lds_after} LDS operations behind the tile stores, the barrier waits lgkmcnt({allowed})")
    if n == 0:
        problems.append("attn_fwd64w_kernel: no counted tile barrier found (kernel renamed or restructured? update acai_omr_amd/_asmcheck.py)")
    return problems, n


_AREG = re.compile(r"\ba(\d+)\b")
_ARANGE = re.compile(r"a\[(\d+):(\d+)\]")


def _regs_va(text):
    """{('v', n), ('a', n)} named in an operand string."""
    used = {("v", r) for r in _regs_of(text)}
    for a, b in _ARANGE.findall(text):
        used |= {("a", r) for r in range(int(a), int(b) + 1)}
    for a in _AREG.findall(text):
        used.add(("a", int(a)))
    return used


def check_asm_mfma_operands(src, kernels=r"_ZN12_GLOBAL__N_1\d+attn_(?:fwd64w|bwd64w_dq|bwd64w_dkv|bwd1p)_kernel(?:ILb[01]EE)?", need=2):
    """Every `v_mfma*` between ;;#ASMSTART / ;;#ASMEND: the VALU instructions within `need` wait states in front of it (s_nop inside the asm
    statement included) must not write its A / B / C source registers."""
    problems, n = [], 0
    for name, body in _functions(src, kernels):
        in_asm = False
        for i, ln in enumerate(body):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not (in_asm and t.startswith("v_mfma")):
                continue
            n += 1
            ops = t.split(None, 1)[1].split(",")
            srcs = set()
            for o in ops[1:]:
                srcs |= _regs_va(o)
            waited = 0
            for j in range(i - 1, -1, -1):
                u = body[j].strip()
                if not u or u.startswith(";") or u.startswith(".") or u.endswith(":"):
                    continue
                mn = re.match(r"s_nop\s+(\d+)", u)
                if mn:
                    waited += int(mn.group(1)) + 1
                elif u.startswith("v_mfma"):
                    waited += 2   # (an accumulate chain on the same C / D registers is legal back to back; an MFMA holds the issue port >= 2 slots)
                elif u.startswith("v_"):
                    dst = u.split(None, 1)[1].split(",")[0] if " " in u else ""
                    if not u.startswith("v_cmp") and _regs_va(dst) & srcs:
                        problems.append(f"{name}: +{i}: `{t}` reads a register written {waited} wait states earlier by `{u}` (needs {need})")
                        break
                    waited += 1
                else:
                    waited += 1
                if waited >= need:
                    break
    if n == 0:
        problems.append("no inline-asm MFMA found (kernels renamed? update acai_omr_amd/_asmcheck.py)")
    return problems, n
