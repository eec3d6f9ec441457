"""Is the cross-attention K/V stream faster when it is already resident in the 256 MB Infinity Cache?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib, ops
dev = "cuda"
B, H, dh, S, L = 8, 16, 64, 4096, 12
L_ = _lib.lib()
kc = [torch.randn(B * H * S * dh, device=dev).to(torch.bfloat16) for _ in range(L)]
vc = [torch.randn(B * H * S * dh, device=dev).to(torch.bfloat16) for _ in range(L)]
off = (torch.arange(B, dtype=torch.int64, device=dev) * (H * S * dh))
lens = torch.full((B,), S, dtype=torch.int32, device=dev)
q = torch.randn(B, H * dh, device=dev)
nsplit = S // 512
partial = torch.empty(B * H * nsplit * (dh + 2), device=dev)
def run(layers, iters=48):
    f = lambda l: _lib.check(L_.acai_decode_attn(q.data_ptr(), q.stride(0), kc[l].data_ptr(), vc[l].data_ptr(), off.data_ptr(), lens.data_ptr(), partial.data_ptr(),
                                                 None, 0, B, H, dh, dh, 512, nsplit, _lib.ACAI_BF16, 0, None, ops._st()), "x")
    for i in range(12): f(layers[i % len(layers)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): f(layers[i % len(layers)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("cycling 12 layers (HBM)      : %.2f us" % run(list(range(12))))
print("one layer repeated (MALL hit): %.2f us" % run([0]))
print("two layers alternating       : %.2f us" % run([0, 1]))
