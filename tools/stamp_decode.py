"""Where does a decode GEMV launch spend its time?  s_memrealtime (100 MHz) stamps inside skinny_mfma_kernel (acai_debug_stamps) over one
captured decode step of the benchmark configuration (8 x 4096-patch memories): per launch, relative to the end of the previous skinny
launch: first / median / last workgroup start, and the median stage boundaries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acai_omr_amd import _lib
from bench import build_model

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
vit = build_model(dev, 8)
lens = [4096] * 8
mem = torch.randn(sum(lens), 1024, device=dev).to(torch.bfloat16)
blocks = vit.decoder.decoder_blocks
blocks.prepare_caches_packed(None, mem, lens)
eng = blocks.engine(dev)
NL = 80
buf = torch.zeros(NL * 1024 * 8, dtype=torch.int64, device=dev)
with torch.cuda.stream(eng.stream):
    eng.arm(eng.B)
    eng.ensure_graph(1)
    eng.arm(eng.B)
    eng.launch_steps(40)
    torch.cuda.synchronize()
    eng.graphs.clear()
    _lib.lib().acai_debug_stamps(buf.data_ptr(), NL)
    g = eng.ensure_graph(1)       # warm-up launch (slots 0..73) + capture (slots advance further; cap stops them)
    _lib.lib().acai_debug_stamps(None, 0)
    # the eager warm-up launch inside ensure_graph wrote slots 0..73: analyse those (a real dependent chain on the stream)
    torch.cuda.synchronize()
st = buf.view(NL, 1024, 8).cpu()
names = ["qkv", "out", "crossq", "crossout", "lin1", "lin2"]
prev_end = None
print("launch            wgs | gap(prev end -> first start) start spread | median stage ns: x+LN  bar1  mfma  bar2  epi | body(med) kernel(total)")
rows = []
for i in range(74):
    s = st[i]
    used = s[:, 0] > 0
    s = s[used].double() * 10.0   # ns
    if s.shape[0] == 0:
        continue
    t0 = s[:, 0]
    first, last_end = float(t0.min()), float(s[:, 5].max())
    med = lambda k: float((s[:, k] - s[:, k - 1]).median())
    name = names[i % 6] if i < 72 else ("unembed" if i == 72 else "?")
    gap = first - prev_end if prev_end is not None else float("nan")
    rows.append((name, s.shape[0], gap, float(t0.max() - t0.min()), med(1), med(2), med(3), med(4), med(5), float((s[:, 5] - s[:, 0]).median()), last_end - first))
    prev_end = last_end
import collections
agg = collections.defaultdict(list)
for r in rows[6:]:
    agg[r[0]].append(r[1:])
for name, v in agg.items():
    t = torch.tensor(v, dtype=torch.float64).nanmean(0)
    print(f"{name:10s} wgs {int(t[0]):4d} | gap {t[1]:7.0f}  spread {t[2]:6.0f} | x+LN {t[3]:6.0f} bar1 {t[4]:5.0f} mfma {t[5]:6.0f} bar2 {t[6]:5.0f} epi {t[7]:5.0f} | body {t[8]:6.0f}  kernel {t[9]:6.0f}")
print("(gap for qkv / crossout spans an attention launch)")
