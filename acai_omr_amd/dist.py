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

    def __init__(self, module, bucket_mb=25.0, group=None):
        self.group = group
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._sync = True
        self._handles = []
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
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
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
            self._launch(bi)

    def _launch(self, bi):
        self._launched[bi] = True
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._handles.append(dist.all_reduce(self.buckets[bi][0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the outstanding all-reduces; buckets the hooks did not complete (a parameter without gradient this step, or the last
        backward ran under no_sync) are reduced here.  A parameter whose .grad is None keeps None: its slice is zeroed first so that stale
        content of an earlier step is not summed into the other ranks' gradients."""
        for bi, (flat, ps) in enumerate(self.buckets):
            if self._launched[bi]:
                continue
            for p in ps:
                if p.grad is None:
                    self._view[id(p)].zero_()
                else:
                    self._attach(p)
            self._launch(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
