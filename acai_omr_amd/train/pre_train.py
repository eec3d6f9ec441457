"""Import-path mirror of `acai_omr/train/pre_train.py` for the names on the hot path: the loop bodies live in `loops.py`."""
from ..config import MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH  # noqa: F401
from .loops import PRETRAIN, save_pretraining_state, set_up_mae  # noqa: F401
from .loops import pretrain_epoch as train_loop  # noqa: F401
from .loops import pretrain_validation as validation_loop  # noqa: F401

EPOCHS, CHECKPOINT_FREQ, BASE_LR, MIN_LR = PRETRAIN["epochs"], PRETRAIN["checkpoint_freq"], PRETRAIN["base_lr"], PRETRAIN["min_lr"]
ADAMW_BETAS, ADAMW_WEIGHT_DECAY, WARMUP_EPOCHS, BATCH_SIZE = PRETRAIN["betas"], PRETRAIN["weight_decay"], PRETRAIN["warmup_epochs"], PRETRAIN["batch_size"]
