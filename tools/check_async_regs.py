"""Command-line form of acai_omr_amd/_asmcheck.py (the build runs the same checks and fails on a hit):
    hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o gemm.s acai_omr_amd/csrc/gemm.hip && python tools/check_async_regs.py gemm.s
    ... -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -S -o w.s acai_omr_amd/csrc/attn_fwd64w.hip && python tools/check_async_regs.py --fwd64w w.s
    ... -S -o b.s acai_omr_amd/csrc/attn_bwd64w.hip && python tools/check_async_regs.py --asm-mfma b.s   (also valid for attn_fwd64w.hip)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acai_omr_amd import _asmcheck  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
src = open(args[0]).read()
check = _asmcheck.check_asm_mfma_operands if "--asm-mfma" in sys.argv else (_asmcheck.check_fwd64w_barrier if "--fwd64w" in sys.argv else _asmcheck.check_untracked_loads)
problems, n = check(src)
for p in problems:
    print(p)
print("checked:", n, "problems:", len(problems))
sys.exit(1 if problems else 0)
