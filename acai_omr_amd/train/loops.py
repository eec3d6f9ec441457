"""The two training loops of the path over the HIP modules: step semantics of
`acai_omr/train/pre_train.py:46-93` (MAE pre-training: fp32, one optimizer step per batch, LR scheduler stepped per EPOCH) and
`acai_omr/train/omr_teacher_force_train.py:59-166` (scheduled-sampling fine-tune: bf16 autocast, gradient accumulation with a flush on the
last batch, LR + teacher-forcing schedulers stepped per OPTIMIZER step, one `StepCounter` tick per optimizer step), the checkpoint layouts of
`pre_train.py:38-44` / `omr_teacher_force_train.py:96-102`, and the hyper-parameters those files fix.

What is left out is what the path does not own: datasets, DataLoader construction, plots / CSV / TensorBoard (a `writer` with `add_scalar` is
used when given).  `dataloader` is anything iterable with `len()` that yields `ragged_collate_fn` batches (lists of (image, target) pairs).

One deliberate difference: the reference reads `loss.item()` every batch, a device sync per step; here the per-batch losses stay on the GPU and
are read once per epoch - the returned averages are the same numbers (fp32 losses summed in double, in batch order)."""
from dataclasses import dataclass

import torch

from ..config import LMX_BOS_TOKEN, LMX_EOS_TOKEN, MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
from ..optim import FusedAdamW
from ..utils import cosine_anneal_with_warmup

# pre_train.py:27-36
PRETRAIN = dict(epochs=500, checkpoint_freq=50, base_lr=1.5e-4, min_lr=1e-6, betas=(0.9, 0.95), weight_decay=0.05, warmup_epochs=50, batch_size=64)
# omr_teacher_force_train.py:30-57
FINE_TUNE = dict(epochs=40, checkpoint_freq=10, fine_tune_base_lr=1e-5, fine_tune_decay_factor=0.9, base_lr=1e-4, min_lr=1e-6, betas=(0.9, 0.95),
                 weight_decay=0.01, warmup_epochs=2, batch_size=8, grad_accumulation_steps=8, encoder_dropout=0.05, transition_head_dropout=0.05,
                 decoder_dropout=0.1, label_smoothing=0.0, initial_tf_prob=1.0, min_tf_prob=0.0, initial_tau=5.0, min_tau=0.1, tf_anneal_epochs=35,
                 soft_epochs=20)


class StepCounter:
    """One global step per optimizer step (acai_omr/utils/utils.py:107-114)."""

    def __init__(self):
        self.global_step = 0

    def increment(self):
        self.global_step += 1


def _on_device(batch, device):
    return [(x.to(device, non_blocking=True), y.to(device, non_blocking=True)) for x, y in batch]


def _mean_of(losses):
    """Average of the per-batch fp32 losses, summed in double in batch order (what `epoch_loss += loss.item()` accumulates)."""
    if not losses:
        return 0.0
    return sum(float(v) for v in torch.stack(losses).cpu().tolist()) / len(losses)


# ---- MAE pre-training -------------------------------------------------------------------------------------------------------------------
def set_up_mae():
    """`set_up_mae` (pre_train.py:156-159)."""
    from ..models.models import MAE
    return MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH)


def set_up_pretrain_optimizer(mae, epochs=PRETRAIN["epochs"], warmup_epochs=PRETRAIN["warmup_epochs"]):
    """AdamW + warm-up / cosine schedule exactly as `pre_train` builds them (pre_train.py:105-107), on the fused optimizer."""
    opt = FusedAdamW(mae.parameters(), lr=PRETRAIN["base_lr"], betas=PRETRAIN["betas"], weight_decay=PRETRAIN["weight_decay"])
    return opt, cosine_anneal_with_warmup(opt, warmup_epochs, epochs, PRETRAIN["min_lr"])


def _dp_scale(ddp, local_count, device):
    """Data parallel: this rank's share of the batch-GLOBAL loss denominator (MAELoss / OMRCELoss divide by a batch-global count,
    models.py:287,788) - scaled local losses + gradient SUM reproduce the single-process global-batch step for ragged shards."""
    from ..dist import global_mean_scale
    return global_mean_scale(float(local_count), group=ddp.group, device=device if torch.device(device).type == "cuda" else None)


def _dp_mean_of(ddp, losses, device):
    """Epoch average of the GLOBAL-batch losses: each entry is this rank's scaled share; one SUM all-reduce of the stacked vector."""
    if not losses:
        return 0.0
    import torch.distributed as dist
    v = torch.stack(losses).double()
    if dist.is_initialized() and dist.get_world_size(ddp.group) > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=ddp.group)
    return sum(v.cpu().tolist()) / len(losses)


def pretrain_epoch(mae, dataloader, loss_fn, optimizer, scheduler, device, ddp=None):
    """`pre_train.train_loop`: every batch is a full step; the scheduler moves once, after the last batch.
    ddp (`dist.GradAllReduce` over `mae`, optional): `dataloader` yields this rank's shard of each global batch; the step is then the
    single-process global-batch step (global-count loss scaling, bucketed gradient SUM all-reduce overlapped with backward)."""
    mae.train()
    losses = []
    if ddp is not None:
        ddp.zero_grad()
    for batch in dataloader:
        pred, loss_mask, target = mae(_on_device(batch, device))
        loss = loss_fn(pred, loss_mask, target)
        if ddp is not None:
            loss = loss * _dp_scale(ddp, loss_mask.sum().item(), device)
        losses.append(loss.detach())
        loss.backward()
        if ddp is not None:
            ddp.finish()
        optimizer.step()
        if ddp is not None:
            ddp.zero_grad()      # in place: gradients stay views of the all-reduce buckets
        else:
            optimizer.zero_grad()
    scheduler.step()
    return _dp_mean_of(ddp, losses, device) if ddp is not None else _mean_of(losses)


def pretrain_validation(mae, dataloader, loss_fn, device):
    """`pre_train.validation_loop`: eval mode (the MAE still masks), no gradients."""
    mae.eval()
    losses = []
    with torch.no_grad():
        for batch in dataloader:
            pred, loss_mask, target = mae(_on_device(batch, device))
            losses.append(loss_fn(pred, loss_mask, target).detach())
    return _mean_of(losses)


def save_pretraining_state(path, mae, optimizer, scheduler):
    """Checkpoint layout of pre_train.py:38-44 (loadable by the reference, and the reference's by `load_training_state`)."""
    torch.save({"mae_state_dict": mae.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "scheduler_state_dict": scheduler.state_dict()}, path)


# ---- scheduled-sampling fine-tune ---------------------------------------------------------------------------------------------------------
@dataclass
class TFConfig:
    tf_prob: float
    tau: float
    use_hard_sampling: bool


class TFScheduler:
    """Teacher-forcing probability (linear) and Gumbel-softmax temperature (geometric) per optimizer step, hard sampling once the soft phase is
    over (omr_teacher_force_train.py:64-83).  `step()` writes the values for the step that FOLLOWS, from the count before the increment."""

    def __init__(self, tf_config, init_tf_prob, min_tf_prob, init_tau, min_tau, soft_epochs, anneal_epochs, num_steps_per_epoch):
        self.tf_config = tf_config
        self.init_tf_prob, self.min_tf_prob = init_tf_prob, min_tf_prob
        self.init_tau, self.min_tau = init_tau, min_tau
        self.soft_steps = soft_epochs * num_steps_per_epoch
        self.anneal_steps = anneal_epochs * num_steps_per_epoch
        self.step_count = 0

    def step(self):
        cfg = self.tf_config
        if self.step_count >= self.soft_steps:
            cfg.use_hard_sampling = True
        progress = self.step_count / self.anneal_steps
        cfg.tf_prob = max(self.init_tf_prob - (self.init_tf_prob - self.min_tf_prob) * progress, self.min_tf_prob)
        cfg.tau = max(self.init_tau * (self.min_tau / self.init_tau) ** progress, self.min_tau)
        self.step_count += 1


def calc_teacher_forcing_prob(epoch, tf_anneal_epochs, initial_prob, min_prob):
    """Per-epoch form (omr_teacher_force_train.py:168-174)."""
    return min_prob if epoch >= tf_anneal_epochs else initial_prob - (initial_prob - min_prob) * (epoch / tf_anneal_epochs)


def calc_tau(epoch, tf_anneal_epochs, initial_tau, min_tau):
    """Per-epoch form (omr_teacher_force_train.py:176-181)."""
    return min_tau if epoch >= tf_anneal_epochs else initial_tau * (min_tau / initial_tau) ** (epoch / tf_anneal_epochs)


class PrepareLMXSequence(torch.nn.Module):
    """LMX string -> token ids with <bos> / <eos> (omr_teacher_force_train.py:85-94); unknown tokens raise KeyError as there."""

    def __init__(self, tokens_to_idxs):
        super().__init__()
        self.tokens_to_idxs = tokens_to_idxs

    def forward(self, lmx: str):
        words = [LMX_BOS_TOKEN, *lmx.strip().split(), LMX_EOS_TOKEN]
        return torch.tensor([self.tokens_to_idxs[w] for w in words])


def set_up_fine_tune_optimizer(vitomr, num_train_batches, epochs=FINE_TUNE["epochs"], warmup_epochs=FINE_TUNE["warmup_epochs"],
                               grad_accumulation_steps=FINE_TUNE["grad_accumulation_steps"]):
    """Layer-wise-LR param groups -> AdamW -> per-optimizer-step warm-up / cosine schedule (omr_teacher_force_train.py:203-212).  Returns
    (optimizer, scheduler, optimizer steps per epoch)."""
    groups, _ = vitomr.create_fine_tune_param_groups(FINE_TUNE["base_lr"], FINE_TUNE["fine_tune_base_lr"], FINE_TUNE["fine_tune_decay_factor"])
    opt = FusedAdamW(groups, betas=FINE_TUNE["betas"], weight_decay=FINE_TUNE["weight_decay"])
    steps = -(num_train_batches // -grad_accumulation_steps)
    return opt, cosine_anneal_with_warmup(opt, warmup_epochs, epochs, FINE_TUNE["min_lr"], num_train_batches=steps), steps


def fine_tune_epoch(vitomr, dataloader, loss_fn, optimizer, scheduler, device, grad_accumulation_steps, tf_config, tf_scheduler, writer=None,
                    counter=None, ddp=None):
    """`omr_teacher_force_train.train_loop`: bf16 autocast forward_train + CE per batch, gradients accumulate (losses are NOT divided by the
    accumulation count, as in the reference), optimizer / LR scheduler / TF scheduler / counter move every `grad_accumulation_steps` batches
    and on the last batch.
    ddp (`dist.GradAllReduce` over `vitomr`, optional): every batch is this rank's shard of a global micro-batch; micro-batches accumulate
    locally (`no_sync`) and the gradients are all-reduced once, by the backward of the micro-batch that precedes the optimizer step."""
    import contextlib
    vitomr.train()
    dev_type = torch.device(device).type
    losses, since_step = [], []
    n = len(dataloader)
    if ddp is not None:
        ddp.zero_grad()
    for i, batch in enumerate(dataloader):
        stepping = (i + 1) % grad_accumulation_steps == 0 or i + 1 == n
        with torch.autocast(device_type=dev_type, dtype=torch.bfloat16):
            pred, target_seqs = vitomr.forward_train(_on_device(batch, device), tf_config.tf_prob, tf_config.tau, tf_config.use_hard_sampling)
            loss = loss_fn(pred, target_seqs)
        if ddp is not None:
            loss = loss * _dp_scale(ddp, (target_seqs != loss_fn.pad_idx).sum().item(), device)
        losses.append(loss.detach().float())
        since_step.append(loss.detach().float())
        with (ddp.no_sync() if (ddp is not None and not stepping) else contextlib.nullcontext()):
            loss.backward()
        if stepping:
            if ddp is not None:
                ddp.finish()
            optimizer.step()
            if ddp is not None:
                ddp.zero_grad()
            else:
                optimizer.zero_grad()
            scheduler.step()
            tf_scheduler.step()
            if writer is not None:
                step = counter.global_step if counter is not None else 0
                writer.add_scalar("train/loss", _mean_of(since_step), step)
                writer.add_scalar("train/hyperparams/base_lr", optimizer.param_groups[0]["lr"], step)
                writer.add_scalar("train/hyperparams/fine_tune_base_lr", optimizer.param_groups[2]["lr"], step)
                writer.add_scalar("train/hyperparams/teacher_forcing_prob", tf_config.tf_prob, step)
                writer.add_scalar("train/hyperparams/tau", tf_config.tau, step)
            since_step = []
            if counter is not None:
                counter.increment()
    return _dp_mean_of(ddp, losses, device) if ddp is not None else _mean_of(losses)


def fine_tune_validation(vitomr, dataloader, loss_fn, device):
    """`omr_teacher_force_train.validation_loop`: eval mode, no gradients, autocast, `forward_eval`."""
    vitomr.eval()
    dev_type = torch.device(device).type
    losses = []
    with torch.no_grad(), torch.autocast(device_type=dev_type, dtype=torch.bfloat16):
        for batch in dataloader:
            pred, target_seqs = vitomr.forward_eval(_on_device(batch, device))
            losses.append(loss_fn(pred, target_seqs).detach())
    return _mean_of(losses)


def save_omr_training_state(path, vitomr, optimizer, scheduler):
    """Checkpoint layout of omr_teacher_force_train.py:96-102."""
    torch.save({"vitomr_state_dict": vitomr.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "scheduler_state_dict": scheduler.state_dict()},
               path)


def load_training_state(path, model, optimizer=None, scheduler=None, map_location=None):
    """Resume from either checkpoint layout (whichever of `mae_state_dict` / `vitomr_state_dict` the file holds)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    key = "mae_state_dict" if "mae_state_dict" in ck else "vitomr_state_dict"
    model.load_state_dict(ck[key])
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck
