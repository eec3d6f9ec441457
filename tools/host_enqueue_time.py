"""How far ahead of the GPU does the host run?  Teacher-forced / MAE training step: seconds until the Python call has ENQUEUED the step
(no synchronisation) against seconds until the GPU has finished it.  python tools/host_enqueue_time.py tf|mae"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "tf"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
orig = bench._timed_steps


def timed(step, steps, dist, dev_, warm=2):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0)
        tot.append(t2 - t0)
    print(f"{which}: host enqueue {min(enq)*1e3:.1f} .. {max(enq)*1e3:.1f} ms, step (enqueue + drain) {min(tot)*1e3:.1f} .. {max(tot)*1e3:.1f} ms", flush=True)
    return orig(step, steps, dist, dev_, warm=0)


bench._timed_steps = timed
if which == "mae":
    bench.bench_mae(dev, 0, 1, None, 32, 512, 2048, 3, "bf16", False)
else:
    bench.bench_tf_step(dev, 16, 512, 2048, 512, 2)
