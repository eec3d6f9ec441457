"""Data parallelism for the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / xGMI.

* Inference / greedy decode shards over independent images: `shard_by_cost` deals a ragged batch to ranks with no collective
  on the tensor path.
* MAE pre-training and the teacher-forced step exchange gradients once per step: `GradAllReduce` keeps every parameter's
  `.grad` as a view into a few flat fp32 buckets (~25 MB, reverse parameter order = the order backward produces them) and
  launches one asynchronous SUM all-reduce per bucket from a post-accumulate-grad hook, so communication overlaps the rest of
  backward; xGMI is point-to-point, so fewer, larger ring all-reduces are the right shape.
* Exactness: both losses divide by a batch-GLOBAL count (masked patches, models.py:287; non-pad tokens, models.py:788).
  `global_mean_scale` all-reduces the local count and returns local/global, so that scaled local losses + gradient SUM
  reproduce the single-process global-batch gradient for ragged shards.  The reference's gradient accumulation sums
  un-normalised micro-batch losses (omr_teacher_force_train.py:117-128): `no_sync()` defers the all-reduce to the last one.
"""
import contextlib

import torch
import torch.distributed as dist


def shard_by_cost(costs, world):
    """Greedy longest-processing-time partition: returns `world` lists of item indices with near-equal total cost."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads, parts = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (loads[j], j))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def global_mean_scale(local_count, group=None, device=None):
    """local_count / sum over ranks of local_count (a python float); one scalar all-reduce."""
    t = torch.tensor([float(local_count)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    total = float(t.item())
    return float(local_count) / total if total > 0 else 0.0


class GradAllReduce:
    """Bucketed gradient SUM all-reduce.  Usage per optimizer step:

        ddp.zero_grad()                     # in place: the buckets stay attached (optimizer.zero_grad() also works, see _hook)
        with ddp.no_sync(): ...backward()   # optional earlier micro-batches: gradients accumulate locally
        loss.backward()                     # hooks launch one asynchronous all-reduce per completed bucket
        ddp.finish()                        # waits; reduces buckets the hooks did not complete
        optimizer.step()

    A second synchronising backward before finish() would reduce a bucket twice - that is refused, loudly."""

    def __init__(self, module, bucket_mb=25.0, group=None, comm_dtype=torch.float32):
        """comm_dtype: what travels over xGMI.  float32 (default): the buckets themselves.  bfloat16: each bucket is cast into a bf16 staging
        buffer on the compute stream, that buffer is all-reduced (half the bytes per ring step: 270 instead of 541 MB for the MAE's 135 M
        parameters) and widened back into the fp32 bucket in finish(); gradients still ACCUMULATE in fp32 locally, only the cross-rank sum is
        rounded (relative error ~2^-9 per addend)."""
        assert comm_dtype in (torch.float32, torch.bfloat16)
        self.group = group
        self.comm_dtype = comm_dtype
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._sync = True
        self._handles = []
        self._staged = []   # (bucket index, bf16 staging buffer) of the launches in flight
        cap = int(bucket_mb * 1024 * 1024 // 4)
        self.buckets = []  # (flat tensor, [params])
        self._view = {}    # id(p) -> its slice of the bucket
        cur, n = [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > cap:
                self._make_bucket(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self._make_bucket(cur)
        # "Some rank produced a gradient for parameter i" rides at the END of the last bucket (launched last: index order), one float per
        # parameter, summed with the gradients - no collective of its own.  finish() reads it only on a rank that has gradient-less parameters.
        flat, ps = self.buckets[-1]
        grown = torch.zeros(flat.numel() + len(self.params), dtype=torch.float32, device=flat.device)
        o = 0
        for p in ps:
            v = grown[o:o + p.numel()].view_as(p)
            self._view[id(p)] = v
            p.grad = v
            o += p.numel()
        self.buckets[-1] = (grown, ps)
        self._flags = grown[o:]
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._bucket_of = {}
        for bi, (_, ps) in enumerate(self.buckets):
            for p in ps:
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)

    def _make_bucket(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
        o = 0
        for p in ps:
            v = flat[o:o + p.numel()].view_as(p)
            self._view[id(p)] = v
            p.grad = v   # autograd accumulates in place into the bucket
            o += p.numel()
        self.buckets.append((flat, ps))

    def zero_grad(self):
        """Zero the buckets in place and (re-)attach every .grad to its slice."""
        for flat, ps in self.buckets:
            flat.zero_()
            for p in ps:
                if p.grad is not self._view[id(p)]:
                    p.grad = self._view[id(p)]
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0

    @contextlib.contextmanager
    def no_sync(self):
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _attach(self, p):
        """`optimizer.zero_grad()` defaults to set_to_none=True, which drops the bucket views; the next backward then allocates a fresh
        .grad.  Move it into the bucket slice (whose old content belongs to a previous step) and re-attach, so the all-reduce sees it."""
        v = self._view[id(p)]
        g = p.grad
        if g is not None and g.data_ptr() != v.data_ptr():
            v.copy_(g)
            p.grad = v

    def _hook(self, p):
        self._attach(p)
        if not self._sync:
            return
        bi = self._bucket_of[id(p)]
        if self._launched[bi]:
            raise RuntimeError("GradAllReduce: a bucket that was already all-reduced received more gradient before finish(); wrap the earlier "
                               "micro-batches of an accumulation in no_sync() and call finish() before optimizer.step()")
        self._pending[bi] += 1
        if self._pending[bi] == len(self.buckets[bi][1]):
            self._ready[bi] = True
            self._launch_ready()

    def _launch_ready(self):
        """Collectives are matched across ranks by CALL ORDER: buckets are therefore launched strictly in index order (bucket i only after
        0 .. i-1), whatever order backward completed them in - a rank whose shard leaves some parameter without gradient completes its buckets
        in a different order than its peers (found by the two-rank test with one rank's branch unused: equal-sized buckets were summed crosswise)."""
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi, have=None):
        self._launched[bi] = True
        if bi == len(self.buckets) - 1:
            # launched from a hook: every parameter of every bucket fired its hook on this rank (all ones, a device-side fill); launched from
            # finish(): `have` lists what this rank has (a small host-to-device copy, on the rare path only)
            if have is None:
                self._flags.fill_(1.0)
            else:
                self._flags.copy_(torch.tensor(have, dtype=torch.float32))
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            flat = self.buckets[bi][0]
            if self.comm_dtype == torch.float32:
                self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                st = flat.to(self.comm_dtype)     # on the compute stream, in front of the collective
                self._staged.append((bi, st))
                self._handles.append(dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the outstanding all-reduces; buckets the hooks did not complete (a parameter without gradient this step, or the last
        backward ran under no_sync) are reduced here.  A parameter whose .grad is None on THIS rank (its shard did not touch it, after a
        zero_grad(set_to_none=True)) contributes zeros - its slice is zeroed first, so stale content of an earlier step is not summed in - and
        afterwards RECEIVES the reduced slice as its .grad IF some other rank produced a gradient for it: otherwise the ranks that did would
        take an AdamW step (weight decay, step count) that this rank skips, and the replicas would drift apart silently.  A parameter NO rank
        produced a gradient for keeps .grad = None on every rank (the per-parameter flags summed at the end of the last bucket say which)."""
        missing = []
        for bi, (flat, ps) in enumerate(self.buckets):
            if self._launched[bi]:
                continue
            for p in ps:
                if p.grad is None:
                    self._view[id(p)].zero_()
                    missing.append(p)
                else:
                    self._attach(p)
            self._ready[bi] = True
        have = None
        if not self._launched[-1]:
            gone = {id(p) for p in missing}
            have = [0.0 if id(p) in gone else 1.0 for p in self.params]
        while self._next < len(self.buckets) and self._ready[self._next]:     # (in index order, as the hooks launch them)
            self._launch(self._next, have if self._next == len(self.buckets) - 1 else None)
            self._next += 1
        for h in self._handles:
            h.wait()
        for bi, st in self._staged:
            self.buckets[bi][0].copy_(st)     # widen the summed bf16 buffer back into the fp32 bucket the .grad views alias
        self._staged = []
        if missing and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # a parameter some OTHER rank produced a gradient for receives the reduced slice (AdamW must step it here too); one that no rank
            # touched keeps .grad = None, as in the single-process global-batch step (and as DistributedDataParallel's used-parameter bitmap does)
            flags = self._flags.cpu()
            for p in missing:
                if float(flags[self._index[id(p)]]) > 0.0:
                    p.grad = self._view[id(p)]
        self._handles = []
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0


# ---- self-check of the data-parallel step on the REAL path (HIP autograd Functions + GradAllReduce + FusedAdamW) ---------------------------
def dp_parity_check(golden_dir, vocab_path, device, group=None):
    """Every rank trains the tiny MAE and the tiny teacher-forced ViTOMR of tests/golden/{mae_small,tf_small}.pt on ITS ragged shard of a
    global batch (global-count loss scaling, bucketed gradient all-reduce, `no_sync` accumulation for the teacher-forced step, fused AdamW)
    and, beside it, the single-process step on the whole global batch; returns the largest |difference| of gradients (relative to the
    tensor's largest gradient) and of the AdamW-updated parameters (relative to lr, on the elements whose gradient is not ~0: AdamW's
    normalised update turns rounding noise on a ~0 gradient into a +-lr move on either side) - fp32: ~1e-5.  Called by `bench.py --gpus N` (field `dp_parity_max_abs_diff`)
    and by the 2-rank GPU test; needs an initialised process group (or none: world size 1)."""
    import os

    from .models.models import MAE, FineTuneOMREncoder, MAELoss, OMRCELoss, OMRDecoder, TeacherForcedViTOMR
    from .optim import FusedAdamW
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = torch.device(device)
    worst = 0.0

    def compare(m_dp, m_ref):
        nonlocal worst
        for (n, a), (_, b) in zip(m_dp.named_parameters(), m_ref.named_parameters()):
            if b.grad is None:
                continue
            worst = max(worst, float((a.grad - b.grad).abs().max()) / max(1e-12, float(b.grad.abs().max())))

    def compare_params(m_dp, m_ref, lr):
        nonlocal worst
        for (n, a), (_, b) in zip(m_dp.named_parameters(), m_ref.named_parameters()):
            if b.grad is None:
                continue
            solid = b.grad.abs() > 1e-3 * b.grad.abs().max()
            if bool(solid.any()):
                worst = max(worst, float((a.detach() - b.detach()).abs()[solid].max()) / lr)

    # ---- MAE: 2 * world images (the fixture's three, cycled, each with its own noise), dealt by patch count
    fx = torch.load(os.path.join(golden_dir, "mae_small.pt"), map_location="cpu", weights_only=False)
    cfg = fx["cfg"]
    n_items = 2 * world + 1
    items = [(fx["imgs"][i % 3].to(dev), fx["tgts"][i % 3].to(dev), fx["noises"][i % 3].roll(i // 3)) for i in range(n_items)]
    costs = [it[0].shape[-1] * it[0].shape[-2] for it in items]
    mine = shard_by_cost(costs, world)[rank]

    def build_mae():
        m = MAE(cfg["mask_ratio"], cfg["P"], cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"], decoder_hidden_dim=cfg["dec_dim"],
                encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"])
        m.load_state_dict(fx["state_dict"])
        return m.to(dev).train()

    ref, dp = build_mae(), build_mae()
    pred, lm, tgt = ref([(a, b) for a, b, _ in items], noises=[c for _, _, c in items])
    MAELoss()(pred, lm, tgt).backward()
    ddp = GradAllReduce(dp, bucket_mb=0.01, group=group)
    ddp.zero_grad()
    pred, lm, tgt = dp([(items[i][0], items[i][1]) for i in mine], noises=[items[i][2] for i in mine])
    (MAELoss()(pred, lm, tgt) * global_mean_scale(float(lm.sum().item()), group=group, device=dev)).backward()
    ddp.finish()
    compare(dp, ref)
    for m in (ref, dp):
        FusedAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05).step()
    compare_params(dp, ref, 1e-3)

    # ---- teacher-forced ViTOMR: two micro-batches accumulated (summed losses, omr_teacher_force_train.py:117-128), one all-reduce
    fx = torch.load(os.path.join(golden_dir, "tf_small.pt"), map_location="cpu", weights_only=False)
    cfg = fx["cfg"]

    def build_tf():
        enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                                 num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"], transformer_dropout=0.0)
        dec = OMRDecoder(cfg["max_len"], vocab_path, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"],
                         mlp_dim=cfg["dec_mlp"], transformer_dropout=0.0)
        m = TeacherForcedViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"], transition_head_dropout=0.0)
        m.load_state_dict(fx["state_dict"])
        return m.to(dev).train()

    ref, dp = build_tf(), build_tf()
    ce = OMRCELoss(ref.decoder.pad_idx)
    micro = []
    for mb in range(2):
        its = [(fx["imgs"][(i + mb) % 3].to(dev), fx["lmx"][(i + 2 * mb) % 3].to(dev)) for i in range(2 * world + 1 - mb)]
        micro.append(its)
    for its in micro:
        pred, tgt = ref(its)
        ce(pred, tgt).backward()
    ddp = GradAllReduce(dp, bucket_mb=0.01, group=group)
    ddp.zero_grad()
    for k, its in enumerate(micro):
        sel = shard_by_cost([it[0].shape[-1] * it[0].shape[-2] + 64 * it[1].numel() for it in its], world)[rank]
        pred, tgt = dp([its[i] for i in sel])
        loss = ce(pred, tgt) * global_mean_scale(float((tgt != ref.decoder.pad_idx).sum().item()), group=group, device=dev)
        if k + 1 < len(micro):
            with ddp.no_sync():
                loss.backward()
        else:
            loss.backward()
    ddp.finish()
    compare(dp, ref)
    t = torch.tensor([worst], dtype=torch.float64, device=dev)
    if dist.is_initialized() and world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
