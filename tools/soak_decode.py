"""Determinism / soak check of the graph-replayed decode: N repetitions of a long greedy decode on the full-size model must give identical
token ids and log-probs (the in-launch split merge and the device-side loop state hold no run-to-run freedom).  python tools/soak_decode.py [reps] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
vit = bench.build_model(dev, 8)
blocks = vit.decoder.decoder_blocks
g = torch.Generator().manual_seed(0)
lens = [4096, 3000, 4096, 1024, 777, 4096, 2048, 4096]
mem = torch.randn(sum(lens), vit.decoder.hidden_dim, generator=g).to(dev)
blocks.prepare_caches_packed(mem, None, lens)
eng = blocks.engine(dev)
ref = None
t0 = time.perf_counter()
for r in range(reps):
    seqs, lps, done = eng.greedy(steps + 1, poll=steps)
    torch.cuda.synchronize()
    cur = (seqs.clone(), lps.clone())
    if ref is None:
        ref = cur
    else:
        assert torch.equal(cur[0], ref[0]), f"repetition {r}: token ids differ"
        assert torch.equal(cur[1], ref[1]), f"repetition {r}: log-probs differ"
    print(f"rep {r}: {done} steps ok", flush=True)
print(f"{reps} x {steps} steps identical; {reps * steps * len(lens) / (time.perf_counter() - t0):.0f} tokens/s incl. host overhead")
