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


def _worker(rank, world, port, out, comm="float32"):
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import GradAllReduce, global_mean_scale
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(10, 6, generator=g)
    M = (torch.rand(10, generator=g) > 0.4).float()
    shard = slice(0, 7) if rank == 0 else slice(7, 10)   # ragged shards: 7 vs 3 rows, different masked counts
    model = _model()
    ddp = GradAllReduce(model, bucket_mb=0.0002, comm_dtype=getattr(torch, comm))          # tiny buckets -> several all-reduces
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


import pytest  # noqa: E402


@pytest.mark.parametrize("comm", ["float32", "bfloat16"])
def test_grad_allreduce_matches_single_process_global_batch(tmp_path, comm):
    """comm = bfloat16: the buckets travel as bf16 (half the bytes over xGMI) and are widened back into the fp32 buckets - the cross-rank sum is
    rounded to bf16, local accumulation (the no_sync micro-batches) stays fp32."""
    out = str(tmp_path / "g.pt")
    port = 29500 + os.getpid() % 2000 + (7 if comm == "bfloat16" else 0)
    mp.spawn(_worker, args=(2, port, out, comm), nprocs=2, join=True)
    g1, g2 = torch.load(out)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(10, 6, generator=g)
    M = (torch.rand(10, generator=g) > 0.4).float()
    model = _model()
    s, c = _masked_mean_loss(model, X, M)
    (s / c).backward()
    ref1 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # bf16 buckets: each rank's addend and the sum are rounded to 8 significant bits
    tol1, tol2 = (1e-6, 1e-5) if comm == "float32" else (2e-2 * float(ref1.abs().max()), None)
    assert torch.allclose(g1, ref1, atol=tol1)          # DP == single-process global batch
    model.zero_grad()
    s, _ = _masked_mean_loss(model, X, M)
    s.backward()
    ref2 = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(g2, ref2, atol=tol2 if tol2 is not None else 2e-2 * float(ref2.abs().max()))          # accumulated sums, one sync
    if comm == "bfloat16":
        assert not torch.equal(g1, ref1)                # (the rounding really happened: the mode is not silently fp32)


def test_shard_by_cost_balances_ragged_batch():
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import shard_by_cost
    costs = [1024, 2048, 2304, 4096, 6144, 6400, 6912, 9216] * 4   # SURVEY config 4 shapes x 4
    parts = shard_by_cost(costs, 8)
    assert sorted(i for p in parts for i in p) == list(range(32))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 1024 and all(len(p) == 4 for p in parts)


# ---- the training loops under data parallelism (acai_omr_amd/train/loops.py with ddp=GradAllReduce) ---------------------------------------
class _ToyMAE(torch.nn.Module):
    """Stands in for the HIP MAE on the CPU: same call contract (batch of (image, target) -> pred, loss_mask, target), ragged batches."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.net = torch.nn.Sequential(torch.nn.Linear(8, 12), torch.nn.GELU(), torch.nn.Linear(12, 8))
        self.unused = torch.nn.Parameter(torch.zeros(3))   # never receives a gradient: its bucket is reduced by finish()

    def forward(self, batch):
        xs = torch.stack([x for x, _ in batch])
        mask = torch.stack([y for _, y in batch])
        return self.net(xs), mask, xs


def _toy_mae_loss(pred, loss_mask, target):
    per = ((pred - target) ** 2).mean(-1)
    return (per * loss_mask).sum() / loss_mask.sum()


class _ToyTF(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(4)
        self.emb = torch.nn.Embedding(11, 6)
        self.out = torch.nn.Linear(6, 11)

    def forward_train(self, batch, tf_prob, tau, hard):
        seqs = torch.stack([y for _, y in batch])
        with torch.autocast("cpu", enabled=False):   # the loop runs under autocast(bf16): keep the toy exact (bf16 GEMMs round per shard)
            return self.out(self.emb(seqs[:, :-1])), seqs[:, 1:]


class _ToyCE(torch.nn.Module):
    pad_idx = 1

    def forward(self, pred, tgt):
        return torch.nn.functional.cross_entropy(pred.reshape(-1, pred.shape[-1]).float(), tgt.reshape(-1), ignore_index=1)


def _toy_data():
    g = torch.Generator().manual_seed(9)
    mae_batches = [[(torch.randn(8, generator=g), (torch.rand(1, generator=g) > 0.3).float().squeeze(0)) for _ in range(n)] for n in (6, 5, 7)]
    for b in mae_batches:     # at least one masked row per shard
        b[0] = (b[0][0], torch.tensor(1.0))
        b[-1] = (b[-1][0], torch.tensor(1.0))
    tf_batches = []
    for n in (5, 6, 4):
        rows = []
        for _ in range(n):
            s = torch.randint(2, 11, (7,), generator=g)
            s[int(torch.randint(3, 7, (1,), generator=g)):] = 1    # ragged <pad> tails: different non-pad counts per shard
            rows.append((torch.zeros(1), s))
        tf_batches.append(rows)
    return mae_batches, tf_batches


class _L(list):
    pass


def _run_loops(model_mae, model_tf, mae_batches, tf_batches, ddp_mae=None, ddp_tf=None, set_to_none=False):
    from acai_omr_amd.train import loops
    from acai_omr_amd.utils import cosine_anneal_with_warmup
    opt = torch.optim.SGD(model_mae.parameters(), lr=0.1)   # (SGD: AdamW's normalised update amplifies rounding of ~0 gradients)
    sch = cosine_anneal_with_warmup(opt, 1, 4, 1e-6)
    if set_to_none:
        opt.zero_grad(set_to_none=True)      # drops the bucket views: the hooks must re-attach
    avg_mae = loops.pretrain_epoch(model_mae, _L(mae_batches), _toy_mae_loss, opt, sch, "cpu", ddp=ddp_mae)
    opt2 = torch.optim.SGD(model_tf.parameters(), lr=0.1)
    sch2 = cosine_anneal_with_warmup(opt2, 1, 4, 1e-6, num_train_batches=2)
    cfg = loops.TFConfig(1.0, 5.0, False)
    tfs = loops.TFScheduler(cfg, 1.0, 1.0, 5.0, 0.1, 1, 2, 2)
    avg_tf = loops.fine_tune_epoch(model_tf, _L(tf_batches), _ToyCE(), opt2, sch2, "cpu", 2, cfg, tfs, ddp=ddp_tf)
    return avg_mae, avg_tf


def _loop_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import GradAllReduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mae_batches, tf_batches = _toy_data()
    cut = lambda b: b[: len(b) // 2 + 1] if rank == 0 else b[len(b) // 2 + 1:]   # noqa: E731  ragged shards
    m1, m2 = _ToyMAE(), _ToyTF()
    d1, d2 = GradAllReduce(m1, bucket_mb=0.0003), GradAllReduce(m2, bucket_mb=0.0003)
    avg = _run_loops(m1, m2, [cut(b) for b in mae_batches], [cut(b) for b in tf_batches], d1, d2, set_to_none=True)
    # a second synchronising backward without finish() must be refused, not silently double-reduced
    d1.zero_grad()
    pred, lm, tgt = m1(cut(mae_batches[0]))
    _toy_mae_loss(pred, lm, tgt).backward()
    try:
        pred, lm, tgt = m1(cut(mae_batches[0]))
        _toy_mae_loss(pred, lm, tgt).backward()
        refused = False
    except RuntimeError:
        refused = True
    d1.finish()
    if rank == 0:
        torch.save((avg, [p.detach().clone() for p in m1.parameters()], [p.detach().clone() for p in m2.parameters()], refused), out)
    dist.barrier()
    dist.destroy_process_group()


def test_training_loops_under_two_ranks_match_single_process(tmp_path):
    """pretrain_epoch / fine_tune_epoch with ddp=GradAllReduce on 2 gloo ranks and ragged shards == the same loops on the whole batches in
    one process: losses and parameters after the epoch (per-batch steps; accumulation of 2 with a flush), with an optimizer.zero_grad(
    set_to_none=True) beforehand (the ADVICE case: dropped bucket views must be re-attached by the hooks)."""
    out = str(tmp_path / "l.pt")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_loop_worker, args=(2, port, out), nprocs=2, join=True)
    avg, p1, p2, refused = torch.load(out)
    sys.path.insert(0, ROOT)
    mae_batches, tf_batches = _toy_data()
    m1, m2 = _ToyMAE(), _ToyTF()
    ref = _run_loops(m1, m2, mae_batches, tf_batches)
    assert refused
    assert abs(avg[0] - ref[0]) < 1e-5 and abs(avg[1] - ref[1]) < 1e-5, (avg, ref)
    for a, b in zip(p1, m1.parameters()):
        assert torch.allclose(a, b.detach(), atol=2e-5), float((a - b.detach()).abs().max())
    for a, b in zip(p2, m2.parameters()):
        assert torch.allclose(a, b.detach(), atol=2e-5), float((a - b.detach()).abs().max())


def _worker_unused(rank, world, port, out):
    """Rank 1's shard never touches the second branch of the model: after optimizer.zero_grad(set_to_none=True) its .grad there is None."""
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import GradAllReduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    model = torch.nn.ModuleList([a, b])
    ddp = GradAllReduce(model, bucket_mb=0.00005)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.1)
    x = torch.ones(3, 4) * (rank + 1)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        y = a(x).sum() + (b(x).sum() if rank == 0 else 0.0)
        y.backward()
        ddp.finish()
        assert all(p.grad is not None for p in model.parameters()), "every rank must step every parameter"
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_without_gradient_on_one_rank_stays_in_step(tmp_path):
    """a parameter whose .grad is None on one rank received zeros in the all-reduce but kept None, so that rank skipped the AdamW
    step (weight decay, step count) the others took.  finish() now hands it the reduced slice: the replicas stay identical."""
    out = str(tmp_path / "u.pt")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_unused, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out)
    assert torch.equal(r0, r1)


def _worker_unused_everywhere(rank, world, port, out):
    """A parameter NO rank produces a gradient for (a frozen-by-construction branch): it must keep .grad = None and stay out of the AdamW step
    on every rank, exactly as in the single-process global-batch step; one that only rank 0 touches is stepped on both."""
    sys.path.insert(0, ROOT)
    from acai_omr_amd.dist import GradAllReduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    model = torch.nn.ModuleList([a, b, c])
    ddp = GradAllReduce(model, bucket_mb=0.00005)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.1)
    x = torch.ones(3, 4) * (rank + 1)
    c0 = [p.detach().clone() for p in c.parameters()]
    none_ok = True
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        y = a(x).sum() + (b(x).sum() if rank == 0 else 0.0)    # c: never
        y.backward()
        ddp.finish()
        none_ok &= all(p.grad is None for p in c.parameters()) and all(p.grad is not None for p in list(a.parameters()) + list(b.parameters()))
        opt.step()
    untouched = all(torch.equal(p.detach(), q) for p, q in zip(c.parameters(), c0))    # no weight decay, no step count
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = torch.tensor([float(none_ok and untouched)])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        torch.save((gathered, bool(ok.item() > 0)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_without_gradient_on_every_rank_keeps_none(tmp_path):
    """finish() used to hand a globally unused parameter an all-zero .grad, so AdamW applied weight decay / moment decay / a step
    count to it that the single-process step skips.  The per-parameter "some rank had a gradient" flags that ride at the end of the last
    bucket now decide: unused everywhere -> .grad stays None on every rank."""
    out = str(tmp_path / "ue.pt")
    port = 35500 + os.getpid() % 2000
    mp.spawn(_worker_unused_everywhere, args=(2, port, out), nprocs=2, join=True)
    (r0, r1), ok = torch.load(out)
    assert ok and torch.equal(r0, r1)


def test_bench_dry_run_two_ranks():
    """`bench.py --gpus 2 --dry-run` under torch.distributed.run (gloo, CPU): the launcher contract, the leg selection at N > 1 and the JSON
    line's keys, incl. config 5's shard deal - what the driver's first multi-GPU run exercises around the kernels."""
    import json
    import subprocess
    port = 33500 + os.getpid() % 2000
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout      # ONE line, from rank 0
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "mae", "config5"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["dry_run"] is True and "workload" in d["config"]
    assert d["tf_step"] is None and d["ragged_decode"] is None            # single-GPU legs are not run at N > 1
    c5 = d["config5"]
    assert c5["images_global"] == 64 and c5["shard_sizes"] == [32, 32] and abs(c5["count_fractions_sum"] - 1.0) < 1e-12
    assert abs(c5["shard_patches"][0] - c5["shard_patches"][1]) <= 1024   # cost-balanced deal of the ragged shapes
    # the line says what it ran on: backend as torch reports it, the world size it saw, one device entry per rank - and never "RCCL" under gloo
    assert d["dist"]["backend"] == "gloo" and d["dist"]["world_size"] == 2 and len(d["dist"]["devices"]) == 2 and "rehearsal" in d["dist"]["transport"]
    assert "RCCL gradient" not in d["mae"]["includes"] and "gloo gradient all-reduce" in d["mae"]["includes"]
    for leg in ("mae", "tf_step"):
        assert "allreduce_exposed_ms" in c5[leg] and "grad_bytes_on_wire" in c5[leg] and c5[leg]["grad_comm_dtype"] == "fp32"
    # and the single-process form selects the single-GPU legs
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["config5"] is None and d1["end_to_end"] is not None and d1["latency_b1"] is not None and d1["tf_step"] is not None


def test_bench_dry_run_eight_ranks_selects_the_multi_gpu_legs():
    """The driver's N = 8 launch, rehearsed on the CPU (8 gloo ranks): default legs = mae + config5 (asserted inside --dry-run), the config-5 keys
    the scaling record will be read from, the bf16 bucket option on the command line, 8 device entries."""
    import json
    import subprocess
    port = 37500 + os.getpid() % 2000
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run", "--grad-comm-dtype", "bf16"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["dist"]["world_size"] == 8 and len(d["dist"]["devices"]) == 8
    assert d["mae"] is not None and d["config5"] is not None and d["tf_step"] is None and d["end_to_end"] is None
    c5 = d["config5"]
    assert c5["images_global"] == 256 and c5["shard_sizes"] == [32] * 8 and max(c5["shard_patches"]) - min(c5["shard_patches"]) <= 1024
    for leg in ("mae", "tf_step"):
        assert "allreduce_exposed_ms" in c5[leg] and c5[leg]["grad_comm_dtype"] == "bf16"
