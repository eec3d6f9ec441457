"""GRPO rollout decode timing: B images x group_size rollouts, full-size decoder, hipGraph sampling steps.
python tools/bench_rollout.py [images] [group] [steps] [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

imgs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 128
S = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
R = imgs * G
vit = bench.build_model(dev, R)
blocks = vit.decoder.decoder_blocks
g = torch.Generator().manual_seed(0)
mem = torch.randn(imgs * S, vit.decoder.hidden_dim, generator=g).to(dev)
for grouped in (True, False):
    if grouped:
        blocks.prepare_caches_packed(mem, None, [S] * imgs, group_size=G)
    else:
        blocks.prepare_caches_packed(mem.view(imgs, 1, S, -1).expand(-1, G, -1, -1).reshape(R * S, -1).contiguous(), None, [S] * R)
    eng = blocks.engine(dev)
    for mode in ("sample", "greedy"):
        u = torch.rand(R, steps + 1, device=dev)
        run = (lambda: eng.sample(steps + 1, 50, 1.2, uniforms=u, poll=steps)) if mode == "sample" else (lambda: eng.greedy(steps + 1, poll=steps))
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, done = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{'shared' if grouped else 'copied'} cross K/V, {mode}: R={R} ({imgs} images x {G}), S={S}: {dt / done * 1e3:.3f} ms/step, {R * done / dt:.0f} tokens/s", flush=True)
