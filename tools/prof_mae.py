"""MAE training step only (for rocprofv3): python tools/prof_mae.py [batch] [steps] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
torch.cuda.set_device(0)
print(bench.bench_mae(torch.device("cuda", 0), 0, 1, None, b, 512, 2048, st, dt))
