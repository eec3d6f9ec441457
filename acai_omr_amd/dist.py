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
    def __init__(self, module, bucket_mb=25.0, group=None):
        self.group = group
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._sync = True
        self._handles = []
        cap = int(bucket_mb * 1024 * 1024 // 4)
        self.buckets = []  # (flat tensor, [params])
        cur, n = [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > cap:
                self._make_bucket(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self._make_bucket(cur)
        self._pending = [0] * len(self.buckets)
        self._bucket_of = {}
        for bi, (_, ps) in enumerate(self.buckets):
            for p in ps:
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)

    def _make_bucket(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
        o = 0
        for p in ps:
            p.grad = flat[o:o + p.numel()].view_as(p)   # autograd accumulates in place into the bucket
            o += p.numel()
        self.buckets.append((flat, ps))

    def zero_grad(self):
        for flat, _ in self.buckets:
            flat.zero_()
        self._pending = [0] * len(self.buckets)

    @contextlib.contextmanager
    def no_sync(self):
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _hook(self, p):
        if not self._sync:
            return
        bi = self._bucket_of[id(p)]
        self._pending[bi] += 1
        if self._pending[bi] == len(self.buckets[bi][1]):
            self._launch(bi)

    def _launch(self, bi):
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._handles.append(dist.all_reduce(self.buckets[bi][0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the outstanding all-reduces; parameters that received no gradient this step still have to be reduced."""
        if self._sync:
            for bi, n in enumerate(self._pending):
                if n != len(self.buckets[bi][1]):   # bucket not launched by the hooks (some parameter got no gradient)
                    self._launch(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        self._pending = [0] * len(self.buckets)
