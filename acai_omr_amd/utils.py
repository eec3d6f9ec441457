"""Host-side helpers the callers on either side of the hot path use: the detokenise edge of inference and
the learning-rate schedules of the two training loops.  Pure Python / stock torch schedulers; they drive `optim.FusedAdamW` exactly as they
drive `torch.optim.AdamW` (only `param_groups[i]["lr"]` changes)."""
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

from .config import LMX_EOS_TOKEN


def stringify_lmx_seq(lmx_seq, idxs_to_tokens):
    """(T,) tensor of LMX token indices starting with <bos> -> one LMX string without <bos> / a trailing <eos>
    (acai_omr/utils/utils.py:194-202; consumer acai_omr/ui/routes.py:68-86)."""
    toks = [idxs_to_tokens[idx.item()] for idx in lmx_seq]
    if toks[-1] == LMX_EOS_TOKEN:
        toks.pop(-1)
    return " ".join(toks[1:])


def stepwise_cosine_anneal_with_warmup(optimizer, warmup_steps, total_epochs, final_lr, num_steps_per_epoch):
    """Linear warm-up from 0.5 % of the base LR, then cosine annealing to final_lr, stepped per minibatch (utils.py:204-208)."""
    warmup = LinearLR(optimizer, start_factor=5e-3, end_factor=1.0, total_iters=warmup_steps)
    anneal = CosineAnnealingLR(optimizer, T_max=total_epochs * num_steps_per_epoch - warmup_steps, eta_min=final_lr)
    return SequentialLR(optimizer, schedulers=[warmup, anneal], milestones=[warmup_steps])


def cosine_anneal_with_warmup(optimizer, warmup_epochs, total_epochs, final_lr, num_train_batches=None):
    """The schedule of pre_train.py:107 (per epoch) and omr_teacher_force_train.py:210 (per minibatch when num_train_batches is given)
    (utils.py:212-222)."""
    if not num_train_batches:
        warmup = LinearLR(optimizer, start_factor=5e-3, end_factor=1.0, total_iters=warmup_epochs)
        anneal = CosineAnnealingLR(optimizer, T_max=total_epochs - warmup_epochs, eta_min=final_lr)
        return SequentialLR(optimizer, schedulers=[warmup, anneal], milestones=[warmup_epochs])
    warm = warmup_epochs * num_train_batches
    warmup = LinearLR(optimizer, start_factor=5e-3, end_factor=1.0, total_iters=warm)
    anneal = CosineAnnealingLR(optimizer, T_max=(total_epochs - warmup_epochs) * num_train_batches, eta_min=final_lr)
    return SequentialLR(optimizer, schedulers=[warmup, anneal], milestones=[warm])


def ragged_collate_fn(batch):
    """DataLoader collate for ragged (image, target) examples: the model layer packs them itself (utils.py:225-229)."""
    return list(batch)
