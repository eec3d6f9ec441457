"""Import-path mirror of `acai_omr/train/omr_teacher_force_train.py` for the names on the hot path: the loop bodies live in `loops.py`,
model construction in `inference/vitomr_inference.py`."""
from ..config import ENCODER_FINE_TUNE_DEPTH, LMX_VOCAB_PATH, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH  # noqa: F401
from ..inference.vitomr_inference import set_up_omr_inference  # noqa: F401
from .loops import (FINE_TUNE, PrepareLMXSequence, TFConfig, TFScheduler, calc_tau, calc_teacher_forcing_prob,  # noqa: F401
                    save_omr_training_state)
from .loops import fine_tune_epoch as train_loop  # noqa: F401
from .loops import fine_tune_validation as validation_loop  # noqa: F401

EPOCHS, CHECKPOINT_FREQ, BASE_LR, FINE_TUNE_BASE_LR = FINE_TUNE["epochs"], FINE_TUNE["checkpoint_freq"], FINE_TUNE["base_lr"], FINE_TUNE["fine_tune_base_lr"]
FINE_TUNE_DECAY_FACTOR, MIN_LR, ADAMW_BETAS, ADAMW_WEIGHT_DECAY = FINE_TUNE["fine_tune_decay_factor"], FINE_TUNE["min_lr"], FINE_TUNE["betas"], FINE_TUNE["weight_decay"]
WARMUP_EPOCHS, BATCH_SIZE, GRAD_ACCUMULATION_STEPS = FINE_TUNE["warmup_epochs"], FINE_TUNE["batch_size"], FINE_TUNE["grad_accumulation_steps"]
ENCODER_DROPOUT, TRANSITION_HEAD_DROPOUT, DECODER_DROPOUT, LABEL_SMOOTHING = (FINE_TUNE["encoder_dropout"], FINE_TUNE["transition_head_dropout"],
                                                                               FINE_TUNE["decoder_dropout"], FINE_TUNE["label_smoothing"])
INITIAL_TEACHER_FORCING_PROB, MIN_TEACHER_FORCING_PROB, INITIAL_TAU, MIN_TAU = (FINE_TUNE["initial_tf_prob"], FINE_TUNE["min_tf_prob"],
                                                                                FINE_TUNE["initial_tau"], FINE_TUNE["min_tau"])
TF_ANNEAL_EPOCHS, SOFT_EPOCHS = FINE_TUNE["tf_anneal_epochs"], FINE_TUNE["soft_epochs"]
