"""One secondary bench leg alone (for rocprofv3): python3 tools/prof_leg.py mae|tf|ragged"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
which = sys.argv[1]
if which == "mae":
    print(json.dumps(bench.bench_mae(dev, 0, 1, None, 32, 512, 2048, 3, "bf16", False)))
elif which == "tf":
    print(json.dumps(bench.bench_tf_step(dev, 16, 512, 2048, 512, 2)))
else:
    print(json.dumps(bench.bench_ragged_decode(dev, 512)))
