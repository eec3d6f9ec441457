"""Data-parallel step on the real HIP path under N ranks == the single-process global-batch step (acai_omr_amd.dist.dp_parity_check).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dp_parity.py [--backend gloo]

The launcher starts the ranks BEFORE anything touches the GPU.  --backend gloo lets several ranks share one card (rehearsal / the
one-GPU test box: RCCL refuses two ranks on one device); the default "nccl" is RCCL over xGMI, one rank per GPU."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(a.backend)
    from acai_omr_amd.dist import dp_parity_check
    d = dp_parity_check(os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "lmx_vocab.txt"), dev)
    if rank == 0:
        print(json.dumps(dict(dp_parity_max_abs_diff=d, world=world, backend=a.backend)), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    assert d < 1e-4, d


if __name__ == "__main__":
    main()
