"""ctypes binding of the C-ABI shared library (include/acai_omr_hip.h).

The product path has NO CPU fallback: if `libacai_omr_hip.so` is missing or a symbol is absent, importing
the ops raises, loudly.  `build()` compiles the library in-tree for gfx950 with hipcc (cross-compiles
without a GPU); the built .so is git-ignored but travels to the GPU box with the source tree.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p, c_size_t

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("ACAI_OMR_LIB") or os.path.join(CSRC, "libacai_omr_hip.so")   # (override: A/B builds of the same sources, tools/ab_*.sh)
SOURCES = ["gemm.hip", "elementwise.hip", "attn_varlen.hip", "attn_fwd64.hip", "attn_fwd64w.hip", "attn_bwd.hip", "attn_bwd64w.hip", "attn_bwd1p.hip", "train.hip", "decode.hip", "resize.hip"]

ACAI_F32, ACAI_BF16 = 0, 1
GEMM_GELU, GEMM_ROUND_BF16 = 1, 2


class AcaiDecLayer(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "self_in_w", "self_in_b", "self_out_w", "self_out_b", "cross_q_w", "cross_q_b", "cross_out_w", "cross_out_b",
        "lin1_w", "lin1_b", "lin2_w", "lin2_b", "n1_w", "n1_b", "n2_w", "n2_b", "n3_w", "n3_b",
        "k_self", "v_self", "k_cross", "v_cross")]


class AcaiAdamWTensor(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64), ("group", c_int32), ("pad_", c_int32),
                ("bias_c1", c_float), ("bias_c2_sqrt", c_float)]


class AcaiCastEntry(Structure):
    _fields_ = [("src", c_void_p), ("dst16", c_void_p), ("dst16t", c_void_p), ("dst32r", c_void_p), ("rows", c_int32), ("cols", c_int32),
                ("tile0", c_int32), ("pad_", c_int32)]


class AcaiAdamWGroup(Structure):
    _fields_ = [(n, c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "pad0_", "pad1_", "pad2_")]


class AcaiDecoder(Structure):
    _fields_ = [(n, c_int32) for n in (
        "B", "E", "H", "dh", "dhp", "F", "V", "L", "Tmax", "dtype", "flags", "max_len",
        "self_chunk", "cross_chunk", "self_nsplit", "cross_nsplit", "bos", "pad", "eos", "cross_group")] + [
        ("layers", POINTER(AcaiDecLayer))] + [(n, c_void_p) for n in (
            "emb", "pos", "fn_w", "fn_b", "unembed_w", "unembed_b", "cross_off", "cross_len", "seqs", "logprobs",
            "step", "finished", "x", "xn", "qkv", "attn", "proj", "hid", "logits", "partial", "tickets", "stats")]


_SIGNATURES = {
    "acai_version": (c_int, []),
    "acai_last_error": (c_char_p, []),
    "acai_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "acai_gemm_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_void_p]),
    "acai_gemm_nt_ex": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "acai_gemm_set_variant": (c_int, [c_int]),
    "acai_gemm": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                          c_int, c_int, c_int, c_void_p]),
    "acai_cross_kv_prefill": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "acai_patchify": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "acai_resize_bicubic_aa": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "acai_resize_to_patches": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p]),
    "acai_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "acai_attn_varlen_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_uint32, c_int, c_void_p]),
    "acai_attn_varlen_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                     c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_float, c_uint32, c_int, c_void_p]),
    "acai_attn_varlen_bwd_ws": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_float, c_uint32, c_int, c_void_p, c_size_t, c_void_p]),
    "acai_attn_varlen_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int]),
    "acai_dropout_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_uint32, c_int, c_int, c_void_p]),
    "acai_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "acai_gelu_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "acai_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "acai_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "acai_cast_weights": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "acai_gemm_dw": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "acai_colsum": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "acai_scatter_add_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "acai_mae_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "acai_ce_loss": (c_int, [c_void_p, c_int, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "acai_debug_stamps": (c_int, [c_void_p, c_int]),
    "acai_debug_lds_dma_oob": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "acai_pe_interp_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "acai_pe_interp_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "acai_cast_f32_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "acai_skinny_gemm": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p]),
    "acai_skinny_gemm_ex": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acai_decode_attn": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "acai_decode_embed": (c_int, [POINTER(AcaiDecoder), c_void_p]),
    "acai_decode_step": (c_int, [POINTER(AcaiDecoder), c_void_p]),
    "acai_decode_sample_step": (c_int, [POINTER(AcaiDecoder), c_void_p, c_int, c_float, c_void_p]),
    "acai_decode_logits": (c_int, [POINTER(AcaiDecoder), c_void_p, c_int, c_void_p]),
    "acai_decode_hidden": (c_int, [POINTER(AcaiDecoder), c_void_p, c_void_p]),
    "acai_decode_merge_in_launch": (c_int, [c_int, c_int]),
    "acai_graph_begin": (c_int, [c_void_p]),
    "acai_graph_end": (c_int, [c_void_p, POINTER(c_void_p)]),
    "acai_graph_launch": (c_int, [c_void_p, c_void_p]),
    "acai_graph_destroy": (c_int, [c_void_p]),
}

_lib = None


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 over csrc/*.hip -> csrc/libacai_omr_hip.so (in-tree).  One object per source (rebuilt only when the
    source or a shared header is newer), compiled in parallel, then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(os.path.dirname(CSRC), "..", "include", "acai_omr_hip.h")]
    hipcc = "hipcc" if subprocess.run(["which", "hipcc"], capture_output=True).returncode == 0 else "/opt/rocm/bin/hipcc"
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    # The attention kernels are bound by VALU issue, and the SLP vectoriser pairs their per-score multiplies into v_pk_mul_f32, which costs
    # more issue time than the two v_mul_f32 it replaces (PMC: +24 % VALU instructions without it, -9 % wave cycles).
    per_file = {"attn_varlen.hip": ["-fno-slp-vectorize"], "attn_fwd64.hip": ["-fno-slp-vectorize"],
                # (one wave per SIMD: the score MFMAs must write VGPRs, the output accumulators are asm-owned AGPRs - see the file header)
                "attn_fwd64w.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"], "attn_bwd.hip": ["-fno-slp-vectorize"],
                "attn_bwd64w.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
                "attn_bwd1p.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]}
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def stale(out, deps):
        return force or not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps)

    # Sources whose correctness rests on something the compiler does not model (asm loads it cannot see, a counted LDS wait): their generated
    # assembly is checked on every rebuild and a hit FAILS the build (acai_omr_amd/_asmcheck.py; ADVICE r3: the check used to be a manual tool).
    asm_checks = {"gemm.hip": ["check_untracked_loads"], "attn_fwd64w.hip": ["check_fwd64w_barrier", "check_asm_mfma_operands"],
                  "attn_bwd64w.hip": ["check_asm_mfma_operands"], "attn_bwd1p.hip": ["check_asm_mfma_operands"]}
    jobs, to_check = [], []
    for src in srcs:
        base = os.path.basename(src)
        obj = os.path.join(objdir, base + ".o")
        asm = os.path.join(objdir, base + ".s")
        rebuilt = stale(obj, [src] + hdrs)
        if rebuilt:
            jobs.append([hipcc] + flags + per_file.get(base, []) + ["-c", src, "-o", obj])
        if base in asm_checks and (rebuilt or stale(asm, [src] + hdrs)):
            jobs.append([hipcc] + flags + per_file.get(base, []) + ["--cuda-device-only", "-S", src, "-o", asm])
            to_check.append((base, asm))
    if verbose:
        for j in jobs:
            print(" ".join(j))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
            for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed: {' '.join(r.args)}\n{r.stdout}\n{r.stderr}")
    from . import _asmcheck
    for base, asm in to_check:
        problems = []
        for chk in asm_checks[base]:
            pr, n = getattr(_asmcheck, chk)(open(asm).read())
            problems += pr
            if verbose:
                print(f"asm check {base} ({chk}): {n} sites, {len(pr)} problems")
        if problems:
            os.remove(asm)   # (so that the next build checks again instead of trusting a stale pass)
            raise RuntimeError(f"generated code of {base} violates an assumption the kernel relies on:\n" + "\n".join(problems))
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    if jobs or stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


def lib():
    """Load (once) and type the C-ABI library.  Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"acai_omr_amd: HIP extension {LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no CPU fallback for the product path")
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7, the same soname
    # /opt/rocm's carries).  Load torch's copy first so that our NEEDED entry resolves to it; loading ours first would
    # pull in /opt/rocm's runtime and torch would later add a second one (streams / events are not shared between them).
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if L.acai_version() != 1:
        raise RuntimeError("acai_omr_amd: ABI version mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {lib().acai_last_error().decode()}")


def exported_symbols():
    return list(_SIGNATURES)
