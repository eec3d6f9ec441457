"""Experiment (GPU box): does the latency-bound decode chain overlap with itself?  One engine with 8 sequences against two engines with 4
sequences each, replaying their step graphs on two streams at once (separate weight copies: a lower bound, the halves share nothing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.amp import autocast
from bench import build_model

dev = torch.device("cuda:0")
H, W, STEPS, WARM = 512, 2048, 256, 32

def setup(batch, seed):
    m = build_model(dev, batch)
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.rand(1, H, W, generator=g).to(dev) for _ in range(batch)]
    with torch.no_grad():
        lat32, _, lens = m.encoder.forward_packed(imgs)
        with autocast(device_type="cuda", dtype=torch.bfloat16):
            mem = m.transition_head.forward_packed(lat32)
        m.decoder.decoder_blocks.prepare_caches_packed(None, mem, lens)
    eng = m.decoder.decoder_blocks.engine(dev)
    eng.stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(eng.stream):
        eng.arm(eng.B); eng.ensure_graph(1); eng.ensure_graph(eng.STEPS_PER_GRAPH); eng.arm(eng.B)
    torch.cuda.synchronize()
    return m, eng

def run(engs, n):
    for e in engs:
        with torch.cuda.stream(e.stream):
            e.launch_steps(n)

def timed(engs, label, rows):
    run(engs, WARM); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(engs, STEPS); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {dt / STEPS * 1e3:.3f} ms/step  {rows * STEPS / dt:.0f} tok/s", flush=True)

with torch.no_grad():
    m8, e8 = setup(8, 1)
    timed([e8], "one engine x 8 rows", 8)
    del m8, e8; torch.cuda.empty_cache()
    ma, ea = setup(4, 2)
    timed([ea], "one engine x 4 rows", 4)
    mb, eb = setup(4, 3)
    timed([ea, eb], "two engines x 4 rows, two streams", 8)
