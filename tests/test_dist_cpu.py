"""world_size-2 gloo tests (CPU): the data-parallel host logic of acai_omr_amd/dist.py - bucketed gradient SUM all-reduce,
global-count loss scaling for ragged shards, gradient accumulation with deferred sync, cost-balanced sharding."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 5), torch.nn.LayerNorm(5))


def _masked_mean_loss(model, x, mask):
    y = model(x)
    per_row = (y ** 2).mean(dim=-1)
    return (per_row * mask).sum(), mask.sum()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import GradAllReduce, global_mean_scale
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(10, 6, generator=g)
    M = (torch.rand(10, generator=g) > 0.4).float()
    shard = slice(0, 7) if rank == 0 else slice(7, 10)   # ragged shards: 7 vs 3 rows, different masked counts
    model = _model()
    ddp = GradAllReduce(model, bucket_mb=0.0002)          # tiny buckets -> several all-reduces
    assert len(ddp.buckets) >= 3
    # one step, global-count normalisation
    ddp.zero_grad()
    s, c = _masked_mean_loss(model, X[shard], M[shard])
    scale = global_mean_scale(float(c))
    ((s / c) * scale).backward()
    ddp.finish()
    g1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    # two micro-batches accumulated (sum of un-normalised losses), sync only on the last
    ddp.zero_grad()
    with ddp.no_sync():
        s, c = _masked_mean_loss(model, X[shard][:2], M[shard][:2])
        s.backward()
    s, c = _masked_mean_loss(model, X[shard][2:], M[shard][2:])
    s.backward()
    ddp.finish()
    g2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    if rank == 0:
        torch.save((g1, g2), out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_matches_single_process_global_batch(tmp_path):
    out = str(tmp_path / "g.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    g1, g2 = torch.load(out)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(10, 6, generator=g)
    M = (torch.rand(10, generator=g) > 0.4).float()
    model = _model()
    s, c = _masked_mean_loss(model, X, M)
    (s / c).backward()
    ref1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(g1, ref1, atol=1e-6)          # DP == single-process global batch
    model.zero_grad()
    s, _ = _masked_mean_loss(model, X, M)
    s.backward()
    ref2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(g2, ref2, atol=1e-5)          # accumulated sums, one sync


def test_shard_by_cost_balances_ragged_batch():
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import shard_by_cost
    costs = [1024, 2048, 2304, 4096, 6144, 6400, 6912, 9216] * 4   # SURVEY config 4 shapes x 4
    parts = shard_by_cost(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(32))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 1024 and all(len(p) == 4 for p in parts)
