"""Constants the hot path needs from the reference's config modules (acai_omr/config.py:12-14,
acai_omr/__init__.py:10-18, acai_omr/train/pre_train.py:16-22, acai_omr/train/omr_teacher_force_train.py:22-27)."""
from enum import Enum

LMX_BOS_TOKEN = "<bos>"
LMX_EOS_TOKEN = "<eos>"
LMX_PAD_TOKEN = "<pad>"

PATCH_SIZE = 16
MASK_RATIO = 0.75
PE_MAX_HEIGHT = 60
PE_MAX_WIDTH = 200
ENCODER_FINE_TUNE_DEPTH = 12
MAX_LMX_SEQ_LEN = 1536
NUM_DECODER_LAYERS = 12
LMX_VOCAB_PATH = "lmx_vocab.txt"


class InferenceEvent(Enum):
    ENCODING_START = "encoding_start"
    ENCODING_FINISH = "encoding_finish"
    STEP = "step"
    INFERENCE_FINISH = "inference_finish"
    ALL_INFERENCE_FINISH = "all_inference_finish"
