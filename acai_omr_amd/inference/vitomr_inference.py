"""Entry points of acai_omr/inference/vitomr_inference.py:51-86 on the MI355X backend.

Same signatures and return contracts.  The plumbing is the reference's: eval(), no_grad(), encoder OUTSIDE autocast
(fp32), transition head + greedy decode INSIDE autocast(bfloat16) with a bf16 KV cache.  Internally the image latent
stays a packed token stream from the encoder to the decode loop (no pad / unpad round trip)."""
import logging

import torch
from torch.amp import autocast

from ..config import (ENCODER_FINE_TUNE_DEPTH, LMX_VOCAB_PATH, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH,
                      InferenceEvent)
from ..models.models import FineTuneOMREncoder, OMRDecoder, ScheduledSamplingViTOMR, ViTOMR

logger = logging.getLogger(__name__)


def set_up_omr_inference(lmx_vocab_path=LMX_VOCAB_PATH, max_batch_size=32, cache_dtype=torch.bfloat16, device="cuda"):
    """Model construction of omr_teacher_force_train.set_up_omr_inference (:265-284) + the cached-decoder swap of
    vitomr_inference.__main__ (:98).  The torchvision image transform is host preprocessing and out of scope."""
    encoder = FineTuneOMREncoder(PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH, ENCODER_FINE_TUNE_DEPTH)
    decoder = OMRDecoder(MAX_LMX_SEQ_LEN, lmx_vocab_path, num_layers=NUM_DECODER_LAYERS)
    vitomr = ScheduledSamplingViTOMR(encoder, None, decoder)
    vitomr.decoder = vitomr.decoder.to_cached_version(max_batch_size, cache_dtype)
    return vitomr.to(device), device


def _encode(vitomr, img):
    # encoder outside autocast, as the reference (its eval fast path is not autocastable, vitomr_inference.py:63)
    with autocast(device_type="cuda", enabled=False):
        return vitomr.encoder.forward_packed(img)


def inference(vitomr: ViTOMR, img, device, max_inference_len=1536):
    """img: one (1,H,W) tensor or a list of them -> (seqs int64 (B,T'), log_probs fp32 (B,T'), seq_mask bool (B,T'))."""
    vitomr.eval()
    with torch.no_grad():
        lat32, _, lens = _encode(vitomr, img)
        with autocast(device_type=device, dtype=torch.bfloat16):
            mem = vitomr.transition_head.forward_packed(lat32)
            bf = mem.dtype == torch.bfloat16
            return vitomr._greedy_packed(None if bf else mem, mem if bf else None, lens, max_inference_len)


def streamed_inference(img, vitomr: ViTOMR, device, max_inference_len=1536, flush_interval=25):
    vitomr.eval()
    with torch.no_grad():
        yield {"type": InferenceEvent.ENCODING_START.value, "payload": None}
        img_latent, latent_attention_mask = vitomr.encoder(img)
        with autocast(device_type=device, dtype=torch.bfloat16):
            img_latent = vitomr.transition_head(img_latent)
            yield {"type": InferenceEvent.ENCODING_FINISH.value, "payload": None}
            for event in vitomr.streamed_cached_greedy_generate(img_latent, latent_attention_mask, max_len=max_inference_len,
                                                                flush_interval=flush_interval):
                yield event
