"""acai_omr_amd: MI355X (gfx950) backend for the acai-omr model hot path (ViT encoder, MAE, KV-cached LMX decoder).

`acai_omr_amd.models.models` / `acai_omr_amd.models.kv_caching` mirror `acai_omr.models.*`;
`acai_omr_amd.inference.vitomr_inference` mirrors the `inference` / `streamed_inference` entry points.
The compute lives in csrc/ (hand-written HIP behind the C ABI of include/acai_omr_hip.h)."""
from .config import InferenceEvent  # noqa: F401

__all__ = ["InferenceEvent"]
