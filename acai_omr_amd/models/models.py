"""MI355X backend mirror of `acai_omr.models.models` (reference: acai_omr/models/models.py).

Same class names, constructor arguments and defaults, method names, return contracts, exception types/messages and
state_dict keys as the reference, so it drops in as the model backend (`load_state_dict` of reference checkpoints
works unchanged).  `nn.TransformerEncoder` / `nn.TransformerDecoder` instances are kept as PARAMETER CONTAINERS
(that is what makes the state_dict keys identical); their `forward` is never called.  All arithmetic runs in the HIP
library on a packed token stream (`engine.py`, `ops.py`); padded `(B, L_max, E)` tensors and bool masks only exist at
the API edge.  There is no CPU fallback: inputs are moved to the parameters' GPU device, and a missing HIP library
raises.

Precision: as in the reference, modules compute in fp32 unless called under `torch.autocast("cuda", bfloat16)`
(vitomr_inference.py:81-84 runs the encoder outside and the head + decoder inside autocast); the cached decoder
follows its cache dtype (vitomr_inference.py:94).
"""
import re

import torch
import torch.nn.functional as F
from torch import nn

from .. import engine as EG
from .. import ops
from ..config import LMX_BOS_TOKEN, LMX_EOS_TOKEN, LMX_PAD_TOKEN, InferenceEvent
from .kv_caching import CachedTransformerDecoder, CachedTransformerDecoderLayer, _wc

NUM_CHANNELS = 1  # grayscale sheet music


def _autocast_prec():
    return "bf16" if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16 else "fp32"


def _training_path_needed(module):
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


def _as_image_list(x, device):
    """The reference iterates `for t in x`: a list/tuple of (1,H,W) tensors, or a bare (1,H,W) tensor whose single
    channel is then iterated (vitomr_inference.py:81 passes the latter)."""
    out = []
    for t in x:
        if t.dim() == 2:
            t = t.unsqueeze(0)
        out.append(t.to(device=device, dtype=torch.float32).contiguous())
    return out


class Encoder(nn.Module):
    """ViT encoder on ragged images (M:14-96): Unfold(P) -> Linear -> + pos_embedding[:h_p,:w_p] -> post-LN blocks."""

    _allow_pe_interpolation = False

    def __init__(self, patch_size, pe_max_height, pe_max_width, num_layers=12, hidden_dim=768, num_heads=12, mlp_dim=3072, transformer_dropout=0.0):
        super().__init__()
        self.patch_size = patch_size
        self.pe_max_height = pe_max_height
        self.pe_max_width = pe_max_width
        self.hidden_dim = hidden_dim
        self.unfold = nn.Unfold(kernel_size=self.patch_size, stride=self.patch_size)  # kept for attribute parity; patchify runs in HIP
        self.pos_embedding = nn.Parameter(torch.zeros(self.pe_max_height, self.pe_max_width, self.hidden_dim))
        nn.init.trunc_normal_(self.pos_embedding, std=0.1)
        self.projection = nn.Linear(in_features=(NUM_CHANNELS * self.patch_size ** 2), out_features=self.hidden_dim)
        self.encoder_blocks = nn.TransformerEncoder(
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.hidden_dim, nhead=num_heads, dim_feedforward=mlp_dim,
                                                     dropout=transformer_dropout, activation="gelu", batch_first=True),
            num_layers=num_layers, norm=nn.LayerNorm(self.hidden_dim, eps=1e-6))

    # ---- host-side helpers ----------------------------------------------------------------------------------------
    def _stacks(self):
        return [self.encoder_blocks]

    def _num_heads(self):
        return self._stacks()[-1].layers[0].self_attn.num_heads

    def _device(self):
        return self.pos_embedding.device

    def _grid(self, t):
        h_p, w_p = t.shape[-2] // self.patch_size, t.shape[-1] // self.patch_size
        if not self._allow_pe_interpolation and (h_p > self.pe_max_height or w_p > self.pe_max_width):
            raise ValueError(f"{h_p} x {w_p} image is too large for max positional embedding grid of shape {self.pe_max_height} x {self.pe_max_width}")
        return h_p, w_p

    def _pe_packed(self, dims, select=None):
        """pos_embedding[:h_p,:w_p].reshape(-1,E) of every image, concatenated (M:50); `select[i]` optionally picks rows
        (MAE keeps ids_keep only, M:123).  One row-gather launch; grids beyond the table are interpolated (OMREncoder)."""
        dev, E = self._device(), self.hidden_dim
        table = self.pos_embedding.detach().reshape(-1, E)
        idx, extra = [], []
        base = table.shape[0]
        for i, (h_p, w_p) in enumerate(dims):
            if h_p > self.pe_max_height or w_p > self.pe_max_width:
                grid = self.interpolate_pe(h_p, w_p).detach().reshape(-1, E)
                rows = torch.arange(base, base + h_p * w_p, dtype=torch.int32)
                base += h_p * w_p
                extra.append(grid)
            else:
                rows = (torch.arange(h_p, dtype=torch.int32).unsqueeze(1) * self.pos_embedding.shape[1] + torch.arange(w_p, dtype=torch.int32).unsqueeze(0)).reshape(-1)
            if select is not None:
                rows = rows[select[i]]
            idx.append(rows)
        if extra:
            table = torch.cat([table] + extra, 0).contiguous()
        return ops.gather_rows(table, torch.cat(idx).to(dev))

    def create_attention_mask(self, seq_lens, max_len):
        arange = torch.arange(end=max_len).unsqueeze(0)
        return arange >= torch.tensor(seq_lens).unsqueeze(1)

    def _prec(self):
        return _autocast_prec()

    def embed_packed(self, x):
        """Packed batchify: returns x32 (M,E), xb (bf16 copy or None), lens, dims.  x: images, or a `utils.PackedPatches` (patch rows written by
        the resize kernel: no Unfold / cast here)."""
        dev = self._device()
        prec = self._prec()
        bf = prec == "bf16"
        P = self.patch_size
        from ..utils import PackedPatches
        if isinstance(x, PackedPatches):
            if x.patch_size != P or x.patches.shape[1] != NUM_CHANNELS * P * P:
                raise ValueError(f"patch rows of size {x.patch_size} for an encoder with patch size {P}")
            dims = list(x.dims)
            for h_p, w_p in dims:
                if not self._allow_pe_interpolation and (h_p > self.pe_max_height or w_p > self.pe_max_width):
                    raise ValueError(f"{h_p} x {w_p} image is too large for max positional embedding grid of shape {self.pe_max_height} x {self.pe_max_width}")
            lens = [h * w for h, w in dims]
            patches = x.patches.to(device=dev)
            want = torch.bfloat16 if bf else torch.float32
            if patches.dtype != want:
                patches = ops.cast_bf16(patches.float().contiguous()) if bf else patches.float()
            patches = patches.contiguous()
        else:
            imgs = _as_image_list(x, dev)
            dims = [self._grid(t) for t in imgs]
            lens = [h * w for h, w in dims]
            patches = torch.empty(sum(lens), NUM_CHANNELS * P * P, dtype=torch.bfloat16 if bf else torch.float32, device=dev)
            r0 = 0
            for t in imgs:
                r0 += ops.patchify(t, P, patches, r0)
        pe = self._pe_packed(dims)
        wc = _wc(self)
        if not isinstance(self.projection, nn.Linear):     # a swapped-in module (the reference's tests use nn.Identity, tests/test_mae.py:12)
            x32 = (self.projection(patches.float()) + pe).contiguous()
            return x32, (ops.cast_bf16(x32) if bf else None), lens, dims
        x32 = ops.gemm_nt(patches, wc.w(self.projection.weight, prec), wc.b(self.projection.bias, prec), residual=pe,
                          out_dtype=torch.float32, round_bf16=bf)
        return x32, (ops.cast_bf16(x32) if bf else None), lens, dims

    def forward_packed(self, x):
        """Encoder on the packed token stream: (x32 (M,E), xb, lens).  This is what the inference entry points use."""
        if _training_path_needed(self):
            from ..train import autograd_path
            return autograd_path.encoder_forward_packed(self, x)
        x32, xb, lens, _ = self.embed_packed(x)
        cu = EG.cu_from_lens(lens, x32.device)
        for st in self._stacks():
            x32, xb = EG.encoder_stack(st, x32, xb, cu, max(lens), self._num_heads(), self._prec(), _wc(self))
        return x32, xb, lens

    def _pad_fill(self):
        # eval + even head count: torch takes the nested-tensor fast path and padded rows leave the stack as zeros, i.e.
        # as the final LayerNorm's bias (torch transformer.py:529-550); otherwise padded rows are unspecified -> zeros.
        st = self._stacks()[-1]
        if not self.training and self._num_heads() % 2 == 0 and st.norm is not None and not torch.is_grad_enabled():
            return st.norm.bias.detach()
        return None

    # ---- reference API ----------------------------------------------------------------------------------------------
    def batchify(self, x):
        x32, _, lens, _ = self.embed_packed(x)
        # the reference pads the patch rows with zeros and THEN projects (M:55-62): a padded row holds projection(0) = the bias
        bias = getattr(self.projection, "bias", None)
        return EG.pad_rows(x32, lens, None if bias is None else bias.detach())

    def forward(self, x):
        x32, _, lens = self.forward_packed(x)
        if x32.requires_grad:
            from ..train import autograd_path
            return autograd_path.pad_rows(x32, lens, self._pad_fill())
        return EG.pad_rows(x32, lens, self._pad_fill())

    def embed_single_image(self, x):
        return self.embed_packed([x])[0].unsqueeze(0)

    def generate(self, x: torch.Tensor):
        return self.forward_packed([x])[0].unsqueeze(0)


class MAEEncoder(Encoder):
    """Encoder whose batchify shuffles and drops `mask_ratio` of each image's patches (M:100-180)."""

    def __init__(self, mask_ratio, patch_size, pe_max_height, pe_max_width, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072):
        super().__init__(patch_size, pe_max_height, pe_max_width, num_layers, hidden_dim, num_heads, mlp_dim, transformer_dropout=0.0)
        self.mask_ratio = mask_ratio

    def mask_ids(self, n, device, noise=None):
        """mask_sequence's index part (M:108-119): noise -> ids_keep, ids_restore, seq_mask (int32, 1 = masked)."""
        len_keep = int(n * (1 - self.mask_ratio))
        if noise is None:
            noise = torch.rand(n, device=device)
        ids_shuffle = torch.argsort(noise)
        ids_restore = torch.argsort(ids_shuffle)
        seq_mask = torch.ones(n, device=noise.device, dtype=torch.int)
        seq_mask[:len_keep] = 0
        return ids_shuffle[:len_keep], ids_restore, seq_mask.index_select(0, ids_restore), len_keep

    def mask_sequence(self, t: torch.Tensor, h_p: int, w_p: int, noise=None):
        """M:106-125 -> (t_masked, pos_embed_slice, unmasked_seq_len, len_keep, seq_mask, ids_restore) for ONE unfolded image
        t (1, C P^2, L).  `noise` (optional, (L,)) injects the masking noise the reference draws with torch.rand (M:110)."""
        from ..train import autograd_path
        return autograd_path.mae_mask_sequence(self, t, h_p, w_p, noise)

    def batchify(self, x, noises=None):
        """M:128-173 -> (embeddings (B, L_keep_max, E), encoder_attention_mask, decoder_attention_mask, kept_seq_lens, unmasked_seq_lens,
        batch_seq_masks (jagged int32), batch_ids_restore (jagged), patchified_dims)."""
        from ..train import autograd_path
        return autograd_path.mae_encoder_batchify(self, x, noises)

    def forward(self, x, noises=None):
        from ..train import autograd_path
        return autograd_path.mae_encoder_forward(self, x, noises)


class MAEDecoder(nn.Module):
    def __init__(self, num_layers=8, hidden_dim=512, num_heads=16, mlp_dim=3072, transformer_dropout=0.0):
        super().__init__()
        self.decoder_blocks = nn.TransformerEncoder(
            encoder_layer=nn.TransformerEncoderLayer(d_model=hidden_dim, nhead=num_heads, dim_feedforward=mlp_dim, dropout=transformer_dropout,
                                                     activation="gelu", batch_first=True),
            num_layers=num_layers, norm=nn.LayerNorm(hidden_dim, eps=1e-6))

    def forward(self, x: torch.Tensor, attention_mask: torch.Tensor):
        from ..train import autograd_path
        return autograd_path.mae_decoder_forward(self, x, attention_mask)


class MAE(nn.Module):
    """Masked auto-encoder (M:197-269).  forward(batch) -> pred (N,L_m,CP^2), loss_mask (N,L_m) bool, target (N,L_m,CP^2)."""

    def __init__(self, mask_ratio, patch_size, pe_max_height, pe_max_width, encoder_hidden_dim=768, decoder_hidden_dim=512,
                 encoder_kwargs={}, decoder_kwargs={}):
        super().__init__()
        self.patch_size = patch_size
        self.encoder = MAEEncoder(mask_ratio, self.patch_size, pe_max_height, pe_max_width, hidden_dim=encoder_hidden_dim, **encoder_kwargs)
        self.decoder_hidden_dim = decoder_hidden_dim
        self.decoder = MAEDecoder(hidden_dim=self.decoder_hidden_dim, **decoder_kwargs)
        self.decoder_embed = nn.Linear(encoder_hidden_dim, self.decoder_hidden_dim)
        self.decoder_unembed = nn.Linear(self.decoder_hidden_dim, NUM_CHANNELS * patch_size ** 2)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.decoder_hidden_dim))
        self.decoder_pos_embedding = nn.Parameter(torch.zeros(pe_max_height, pe_max_width, self.decoder_hidden_dim))
        nn.init.trunc_normal_(self.mask_token, std=0.1)
        nn.init.trunc_normal_(self.decoder_pos_embedding, std=0.1)
        self.unfold = nn.Unfold(kernel_size=self.patch_size, stride=self.patch_size)

    def prepare_for_decoder(self, latent: torch.Tensor, kept_seq_lens, unmasked_seq_lens, batch_ids_restore: torch.Tensor, patchified_dims):
        """M:219-241: per sequence drop the padding, append mask tokens, unshuffle by ids_restore, add the decoder PE slice; returns the
        zero-padded (B, L_max, D) decoder input."""
        from ..train import autograd_path
        return autograd_path.mae_prepare_for_decoder(self, latent, kept_seq_lens, unmasked_seq_lens, batch_ids_restore, patchified_dims)

    def forward(self, batch, noises=None):
        """`noises`: optional list of per-image noise vectors (injected masking noise for parity runs; the reference draws
        torch.rand on the model's device, M:110)."""
        from ..train import autograd_path
        return autograd_path.mae_forward(self, batch, noises)

    def forward_packed(self, batch, noises=None):
        from ..train import autograd_path
        return autograd_path.mae_forward(self, batch, noises, packed=True)


class MAELoss(nn.Module):
    """Per-patch normalised-pixel MSE over masked patches (M:271-288); unbiased variance, eps inside the sqrt."""

    def forward(self, pred, loss_mask, target):
        from ..train import autograd_path
        return autograd_path.mae_loss(pred, loss_mask, target)


class OMREncoder(Encoder):
    """Encoder that bilinearly interpolates the PE grid for images beyond it instead of raising (M:290-332)."""

    _allow_pe_interpolation = True

    def interpolate_pe(self, h_p, w_p):
        """(h_p, w_p, E) bilinear resampling of the PE grid, align_corners=False (M:291-302), by the HIP kernel; differentiable w.r.t.
        pos_embedding when it requires grad (the reference interpolates in batchify in every mode, M:315-318)."""
        from ..train import autograd_path
        return autograd_path.PeInterpFn.apply(self.pos_embedding, h_p, w_p).view(h_p, w_p, self.hidden_dim)


class FineTuneOMREncoder(OMREncoder):
    """Encoder split into `frozen_blocks` (no final norm) and `fine_tune_blocks` (with it) (M:334-376)."""

    def __init__(self, patch_size, pe_max_height, pe_max_width, fine_tune_depth, num_layers=12, hidden_dim=768, num_heads=12, mlp_dim=3072,
                 transformer_dropout=0.05):
        super().__init__(patch_size, pe_max_height, pe_max_width, num_layers, hidden_dim, num_heads, mlp_dim)
        assert fine_tune_depth > 0, "If using FineTuneOMREncoder, fine-tune depth should be at least 1"
        del self.encoder_blocks
        self.fine_tune_depth = fine_tune_depth
        self.num_layers = num_layers
        self.num_frozen_layers = self.num_layers - self.fine_tune_depth
        self.superclass_kwargs = {"num_heads": num_heads, "mlp_dim": mlp_dim, "transformer_dropout": transformer_dropout}
        kw = {"d_model": self.hidden_dim, "nhead": num_heads, "dim_feedforward": mlp_dim, "activation": "gelu", "batch_first": True}
        if self.num_frozen_layers == 0:
            self.frozen_blocks = None
        else:
            self.frozen_blocks = nn.TransformerEncoder(encoder_layer=nn.TransformerEncoderLayer(dropout=0.0, **kw), num_layers=self.num_frozen_layers)
        self.fine_tune_blocks = nn.TransformerEncoder(encoder_layer=nn.TransformerEncoderLayer(dropout=transformer_dropout, **kw),
                                                      num_layers=self.fine_tune_depth, norm=nn.LayerNorm(self.hidden_dim, eps=1e-6))

    def _stacks(self):
        return ([self.frozen_blocks] if self.frozen_blocks is not None else []) + [self.fine_tune_blocks]


class OMRDecoder(nn.Module):
    """Autoregressive LMX decoder (M:378-528): embedding + learned positions + post-LN decoder blocks + unembed."""

    def __init__(self, max_lmx_seq_len, lmx_vocab_path, num_layers=10, hidden_dim=1024, num_heads=16, mlp_dim=4096, transformer_dropout=0.1,
                 use_caching=False, max_batch_size=None, cache_dtype=None):
        super().__init__()
        self.max_lmx_seq_len = max_lmx_seq_len
        self.lmx_vocab_path = lmx_vocab_path
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.num_heads = num_heads
        self.head_dim = hidden_dim / num_heads
        self.mlp_dim = mlp_dim
        self.transformer_dropout = transformer_dropout
        with open(lmx_vocab_path, "r") as f:
            tokens = [line.strip() for line in f if line.strip()]
        self.tokens_to_idxs = {token: i for i, token in enumerate(tokens)}
        self.idxs_to_tokens = {i: token for i, token in enumerate(tokens)}
        self.pad_idx = self.tokens_to_idxs[LMX_PAD_TOKEN]
        self.bos_idx = self.tokens_to_idxs[LMX_BOS_TOKEN]
        self.eos_idx = self.tokens_to_idxs[LMX_EOS_TOKEN]
        self.vocab_size = len(tokens)
        self.vocab_embedding = nn.Embedding(self.vocab_size, self.hidden_dim, padding_idx=self.pad_idx)
        self.pos_embedding = nn.Parameter(torch.zeros(self.max_lmx_seq_len, self.hidden_dim))
        nn.init.trunc_normal_(self.pos_embedding, std=0.1)
        lkw = dict(d_model=self.hidden_dim, nhead=num_heads, dim_feedforward=mlp_dim, dropout=transformer_dropout, activation="gelu", batch_first=True)
        if use_caching:
            self.decoder_blocks = CachedTransformerDecoder(decoder_layer=CachedTransformerDecoderLayer(**lkw), num_layers=num_layers,
                                                           max_batch_size=max_batch_size, max_decoder_seq_len=max_lmx_seq_len,
                                                           cache_dtype=cache_dtype, norm=nn.LayerNorm(self.hidden_dim, eps=1e-6))
            self.decoder_blocks.__dict__["_omr"] = self
        else:
            self.decoder_blocks = nn.TransformerDecoder(decoder_layer=nn.TransformerDecoderLayer(**lkw), num_layers=num_layers,
                                                        norm=nn.LayerNorm(self.hidden_dim, eps=1e-6))
        self.unembed = nn.Linear(self.hidden_dim, self.vocab_size)

    def to_cached_version(self, max_batch_size, cache_dtype):
        return OMRDecoder(self.max_lmx_seq_len, self.lmx_vocab_path, self.num_layers, self.hidden_dim, self.num_heads, self.mlp_dim,
                          self.transformer_dropout, use_caching=True, max_batch_size=max_batch_size, cache_dtype=cache_dtype)

    # ---- teacher-forced / uncached batch paths -----------------------------------------------------------------------------
    def forward_packed(self, inputs, lens_t, mem32, memb, lens_s, token_idxs_input=True, prec=None):
        """Teacher-forced decoder on packed streams.  inputs: packed token ids (sum T,) int or packed embeddings (sum T, E);
        positions restart at 0 in every sequence (M:465-466).  Returns packed logits (sum T, V) fp32."""
        prec = prec or _autocast_prec()
        bf = prec == "bf16"
        dev, E, H = self.pos_embedding.device, self.hidden_dim, self.num_heads
        wc = _wc(self)
        pos_idx = torch.cat([torch.arange(t, dtype=torch.int32) for t in lens_t]).to(dev)
        x32 = ops.gather_rows(self.pos_embedding.detach(), pos_idx)
        if token_idxs_input:
            x32 = ops.gather_rows(self.vocab_embedding.weight.detach(), inputs.to(device=dev, dtype=torch.int32).contiguous(), add=x32)
        else:
            x32 = x32 + inputs.to(dev).float()
        xb = ops.cast_bf16(x32) if bf else None
        cu_t, cu_s = EG.cu_from_lens(lens_t, dev), EG.cu_from_lens(lens_s, dev)
        mem = memb if bf else mem32
        if mem is None:
            mem = ops.cast_bf16(mem32)
        mt, dh = max(lens_t), E // H
        for ly in self.decoder_blocks.layers:
            sa, ca = ly.self_attn, ly.multihead_attn
            qkv = EG.linear(x32, xb, sa.in_proj_weight, sa.in_proj_bias, prec, wc)
            a = ops.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu_t, cu_t, H, dh, mt, causal=True)
            y = ops.gemm_nt(a, wc.w(sa.out_proj.weight, prec), wc.b(sa.out_proj.bias, prec), residual=x32, round_bf16=bf)
            x32, xb = ops.layernorm(y, ly.norm1.weight.detach(), ly.norm1.bias.detach(), ly.norm1.eps, want_bf16=bf)
            cdt = torch.bfloat16 if bf else torch.float32
            q = ops.gemm_nt(xb if bf else x32, wc.w(ca.in_proj_weight, prec)[:E], wc.b(ca.in_proj_bias, prec)[:E], out_dtype=cdt, round_bf16=bf)
            kv = ops.gemm_nt(mem, wc.w(ca.in_proj_weight, prec)[E:], wc.b(ca.in_proj_bias, prec)[E:], out_dtype=cdt, round_bf16=bf)
            a = ops.attn_varlen(q, kv[:, :E], kv[:, E:], cu_t, cu_s, H, dh, mt)
            y = ops.gemm_nt(a, wc.w(ca.out_proj.weight, prec), wc.b(ca.out_proj.bias, prec), residual=x32, round_bf16=bf)
            x32, xb = ops.layernorm(y, ly.norm2.weight.detach(), ly.norm2.bias.detach(), ly.norm2.eps, want_bf16=bf)
            h = EG.linear(x32, xb, ly.linear1.weight, ly.linear1.bias, prec, wc, gelu=True)
            y = ops.gemm_nt(h, wc.w(ly.linear2.weight, prec), wc.b(ly.linear2.bias, prec), residual=x32, round_bf16=bf)
            x32, xb = ops.layernorm(y, ly.norm3.weight.detach(), ly.norm3.bias.detach(), ly.norm3.eps, want_bf16=bf)
        nrm = self.decoder_blocks.norm
        x32, xb = ops.layernorm(x32, nrm.weight.detach(), nrm.bias.detach(), nrm.eps, want_bf16=bf)
        return EG.linear(x32, xb, self.unembed.weight, self.unembed.bias, prec, wc, out_dtype=torch.float32)

    def forward(self, input_seqs, img_latent, lmx_attention_mask, latent_attention_mask, token_idxs_input=True, checkpoint_grads=False):
        """Teacher-forced logits (B, L_lmxmax, V) (M:445-483).  <pad> positions (lmx_attention_mask True) are not computed
        and come back as zeros (the reference leaves unspecified values there; OMRCELoss ignores them)."""
        T = input_seqs.shape[1]
        if T > self.max_lmx_seq_len:
            raise ValueError(f"{T} long lmx sequence length is too long for max sequence length of {self.max_lmx_seq_len}")
        if _training_path_needed(self) or (torch.is_grad_enabled() and (img_latent.requires_grad or (not token_idxs_input and input_seqs.requires_grad))):
            from ..train import autograd_path
            return autograd_path.decoder_forward(self, input_seqs, img_latent, lmx_attention_mask, latent_attention_mask, token_idxs_input, checkpoint_grads)
        dev = self.pos_embedding.device
        B = input_seqs.shape[0]
        lens_t = [T] * B if lmx_attention_mask is None else (~lmx_attention_mask).sum(dim=1).tolist()
        mem32, lens_s = EG.unpad_rows(img_latent.to(dev), latent_attention_mask)
        if token_idxs_input:
            packed_in = torch.cat([input_seqs[b, :l] for b, l in enumerate(lens_t)]).to(dev)
        else:
            packed_in = torch.cat([input_seqs[b, :l] for b, l in enumerate(lens_t)], 0)
        logits = self.forward_packed(packed_in, lens_t, mem32, None, lens_s, token_idxs_input)
        out = torch.zeros(B, T, self.vocab_size, dtype=torch.float32, device=dev)
        o = 0
        for b, l in enumerate(lens_t):
            out[b, :l] = logits[o:o + l]
            o += l
        return out.to(torch.bfloat16) if _autocast_prec() == "bf16" else out

    def generate(self, input_seqs, img_latent, latent_attention_mask=None):
        seq_len = input_seqs.shape[1]
        if seq_len > self.max_lmx_seq_len:
            raise ValueError(f"{seq_len} long lmx sequence length is too long for max sequence length of {self.max_lmx_seq_len}")
        return self.forward(input_seqs, img_latent, None, latent_attention_mask)

    # ---- KV-cached path ------------------------------------------------------------------------------------------------
    def prepare_caches(self, encoder_memory):
        if not isinstance(self.decoder_blocks, CachedTransformerDecoder):
            raise RuntimeError("Trying to use cached inference pathway with an uncached TransformerDecoder instance")
        self.decoder_blocks.prepare_caches(encoder_memory)

    def cached_generate(self, token_t: torch.Tensor, time_step: int, latent_attention_mask=None):
        """Logits (B,1,V) for the token after `token_t` (B,1); pos_embedding is indexed with `time_step` literally."""
        if time_step >= self.max_lmx_seq_len:
            raise RuntimeError(f"{time_step + 1} decoding steps is too long for max sequence length of {self.max_lmx_seq_len}")
        if not isinstance(self.decoder_blocks, CachedTransformerDecoder):
            raise RuntimeError("Trying to use cached inference pathway with an uncached TransformerDecoder instance")
        blocks = self.decoder_blocks
        blocks._materialise(latent_attention_mask)
        eng = blocks.engine(self.pos_embedding.device)
        logits = eng.logits_step(token_t, time_step)
        for c in blocks.self_attn_caches:
            c._pos += 1
        logits = logits.view(-1, 1, self.vocab_size).clone()
        return logits.to(torch.bfloat16) if eng.bf else logits


def batchify_and_split_lmx_seqs(lmx_seqs, pad_idx, device):
    """Pad with <pad>, inputs = seq[:, :-1], targets = seq[:, 1:], mask = inputs == pad (M:531-540).  Integer bookkeeping."""
    B, Lm = len(lmx_seqs), max(int(s.shape[0]) for s in lmx_seqs)
    full = torch.full((B, Lm), pad_idx, dtype=lmx_seqs[0].dtype, device=lmx_seqs[0].device)
    for i, s in enumerate(lmx_seqs):
        full[i, :s.shape[0]] = s
    input_seqs, target_seqs = full[:, :-1], full[:, 1:]
    return input_seqs, target_seqs, (input_seqs == pad_idx).to(device)


class _TransitionHead(nn.Sequential):
    """Linear, GELU, Dropout, Linear (M:655-660) as a parameter container; both GEMMs (bias+GELU fused) run in HIP."""

    def forward_packed(self, x32, xb=None, prec=None):
        prec = prec or _autocast_prec()
        bf = prec == "bf16"
        wc = _wc(self)
        h = EG.linear(x32, xb, self[0].weight, self[0].bias, prec, wc, gelu=True)
        y = EG.linear(None if bf else h, h if bf else None, self[3].weight, self[3].bias, prec, wc)
        return y  # (M, E_dec) in the compute dtype

    def forward(self, x):
        if _training_path_needed(self) or (torch.is_grad_enabled() and x.requires_grad):
            from ..train import autograd_path
            return autograd_path.head_forward(self, x)
        shp = x.shape
        y = self.forward_packed(x.reshape(-1, shp[-1]).float().contiguous())
        return y.view(*shp[:-1], y.shape[-1])


class ViTOMR(nn.Module):
    def __init__(self, encoder, transition_head, decoder):
        super().__init__()
        self.encoder = encoder
        self.transition_head = transition_head
        self.decoder = decoder

    def create_inference_mask(self, seqs):
        """True up to and including each row's first <eos> (M:550-559)."""
        eos_mask = seqs == self.decoder.eos_idx
        seen = eos_mask.int().cumsum(dim=-1)
        return (seen == 0) | (eos_mask & (seen == 1))

    def mask_and_clip_seqs(self, seqs, seq_log_probs):
        seq_mask = self.create_inference_mask(seqs)
        seqs = seqs.masked_fill(~seq_mask, self.decoder.pad_idx)
        seq_log_probs = seq_log_probs.masked_fill(~seq_mask, 0.0)
        n = int(seq_mask.sum(dim=-1).max())
        return seqs[:, :n], seq_log_probs[:, :n], seq_mask[:, :n]

    def cached_set_up_inference(self, img_latent, max_len):
        self.decoder.prepare_caches(img_latent)
        B, dev = img_latent.shape[0], img_latent.device
        seqs = torch.full([B, max_len], fill_value=self.decoder.pad_idx, dtype=torch.long, device=dev)
        seqs[:, 0] = self.decoder.bos_idx
        return seqs, torch.zeros_like(seqs, dtype=torch.float), torch.full([B], fill_value=False)

    def cached_get_next_token(self, seqs, t, latent_attention_mask):
        """argmax + log-prob of the next token (M:575-583); passes `t` as the position of token t-1 (quirk Q1)."""
        logits = self.decoder.cached_generate(seqs[:, t - 1].unsqueeze(1), t, latent_attention_mask).squeeze(1)
        idx = torch.argmax(logits, dim=-1)
        lp = F.log_softmax(logits, dim=-1).gather(-1, idx.unsqueeze(1)).squeeze(1)
        return idx, lp

    def _greedy_packed(self, mem32, memb, lens, max_len, on_chunk=None):
        blocks = self.decoder.decoder_blocks
        if not isinstance(blocks, CachedTransformerDecoder):
            raise RuntimeError("Trying to use cached inference pathway with an uncached TransformerDecoder instance")
        blocks.prepare_caches_packed(mem32, memb, lens)
        eng = blocks.engine(self.decoder.pos_embedding.device)
        seqs, lps, _ = eng.greedy(max_len, on_chunk=on_chunk)
        return self.mask_and_clip_seqs(seqs.clone(), lps.clone())

    def cached_greedy_generate(self, img_latent, latent_attention_mask=None, max_len=1536):
        """Batched greedy decode with KV caching (M:600-615) -> seqs (B,T') int64, log_probs (B,T') fp32, mask (B,T') bool.
        The whole loop runs as replays of one captured hipGraph; the host only polls an "all finished" counter."""
        mem32, lens = EG.unpad_rows(img_latent, latent_attention_mask)
        return self._greedy_packed(mem32, None, lens, max_len)

    def streamed_cached_greedy_generate(self, img_latent, latent_attention_mask=None, max_len=1536, flush_interval=25):
        """Generator of {"type", "payload"} events (M:625-647); single image only."""
        if img_latent.shape[0] != 1:
            raise ValueError("Streamed generation only supports single image batches")
        mem32, lens = EG.unpad_rows(img_latent, latent_attention_mask)
        blocks = self.decoder.decoder_blocks
        if not isinstance(blocks, CachedTransformerDecoder):
            raise RuntimeError("Trying to use cached inference pathway with an uncached TransformerDecoder instance")
        blocks.prepare_caches_packed(mem32, None, lens)
        eng = blocks.engine(self.decoder.pos_embedding.device)
        # replay the decode graph flush_interval tokens at a time; after each chunk hand out the freshly written tokens
        # (the reference yields STEP at every t % flush_interval == 0 that did not finish the sequence - also at t == max_len - 1, M:641-645)
        for t_done, finished in eng.greedy_chunks(max_len, flush_interval):
            if finished:
                break
            if t_done % flush_interval == 0:
                buf = eng.seqs[:1, t_done - flush_interval + 1:t_done + 1].to(torch.int)
                yield {"type": InferenceEvent.STEP.value, "payload": {"tokens": buf}}
        seqs, lps, mask = self.mask_and_clip_seqs(eng.seqs[:1, :max_len].clone(), eng.logprobs[:1, :max_len].clone())
        yield {"type": InferenceEvent.INFERENCE_FINISH.value, "payload": {"sequence": seqs, "log_probs": lps, "mask": mask}}


class GRPOViTOMR(ViTOMR):
    """ViTOMR prepared for GRPO (M:840-1049): the rollout policy - sampling decode with KV caching - and the helpers around it.  The reward
    functions and the GRPO training loop itself stay outside the hot path; the rollout decode is SURVEY 8f-1."""

    def __init__(self, encoder, transition_head, decoder, teacher_forced_state_dict):
        super().__init__(encoder, transition_head, decoder)
        if isinstance(self.encoder, FineTuneOMREncoder):
            teacher_forced_state_dict = self.convert_teacher_forced_state_dict(teacher_forced_state_dict, encoder.num_frozen_layers)
            self.encoder = OMREncoder(self.encoder.patch_size, self.encoder.pe_max_height, self.encoder.pe_max_width, self.encoder.num_layers,
                                      self.encoder.hidden_dim, **encoder.superclass_kwargs)
        self.load_state_dict(teacher_forced_state_dict)
        self.freeze_component(self.encoder)
        self.freeze_component(self.transition_head)

    def freeze_component(self, component):
        for param in component.parameters():
            param.requires_grad = False
        for child in component.modules():
            if isinstance(child, nn.Dropout):
                child.p = 0.0

    def convert_teacher_forced_state_dict(self, teacher_forced_state_dict, num_frozen_layers):
        """frozen_blocks / fine_tune_blocks keys -> one encoder_blocks stack, fine-tune layer numbers shifted by num_frozen_layers (M:860-880)."""
        converted = {}
        pat = re.compile(r"(?:\w|\.)+?(\d+)(?:\w|\.)+")
        for name in teacher_forced_state_dict.keys():
            if "frozen_blocks" in name:
                new = name.replace("frozen_blocks", "encoder_blocks")
            elif "fine_tune_blocks" in name:
                new = name.replace("fine_tune_blocks", "encoder_blocks")
                m = pat.match(name)
                if m:
                    n = int(m.group(1))
                    new = new.replace(f"layers.{n}", f"layers.{n + num_frozen_layers}")
            else:
                new = name
            converted[new] = teacher_forced_state_dict[name]
        return converted

    def create_rollout_mask(self, rollouts):
        """Name the reference's own tests use (tests/test_vitomr.py:438-442) for the mask `cached_forward_rollout_policy` returns: True up to
        and including each row's first <eos> - the same rule as `create_inference_mask`."""
        return self.create_inference_mask(rollouts)

    def expand_img_latent_for_rollout(self, img_latent, latent_attention_mask, group_size):
        img_latent = img_latent.unsqueeze(1).expand(-1, group_size, -1, -1).flatten(start_dim=0, end_dim=1)
        latent_attention_mask = latent_attention_mask.unsqueeze(1).expand(-1, group_size, -1).flatten(start_dim=0, end_dim=1)
        return img_latent, latent_attention_mask

    def uncached_forward_rollout_policy(self, img_latent, latent_attention_mask, max_actions=768, top_k=50, temperature=1.2):
        """The reference's deprecated rollout policy without KV caching (M:897-945; "absurdly slow ... treated as deprecated"): every step
        re-runs `decoder.generate` (the uncached HIP forward) on the whole prefix.  Its arithmetic differs from the cached policy's and is kept:
        log-probs are log_softmax over the FULL vocabulary of the top-k-masked logits divided by the temperature, the outputs are not clipped
        to the longest rollout.  Only the small per-step glue (top-k of 227 logits, the multinomial draw) runs as torch ops."""
        device = img_latent.device
        R = img_latent.shape[0]
        rollouts = torch.full([R, max_actions], fill_value=self.decoder.pad_idx, dtype=torch.long, device=device)
        rollouts[:, 0] = self.decoder.bos_idx
        rollout_log_probs = torch.zeros_like(rollouts, dtype=torch.float, device=device)
        for t in range(1, max_actions):
            logits = self.decoder.generate(rollouts[:, :t], img_latent, latent_attention_mask=latent_attention_mask)[:, -1, :].float()
            top_k_logits, top_k_indices = torch.topk(logits, top_k, dim=-1)
            softmax_logits = torch.full_like(logits, float("-inf")).scatter(-1, top_k_indices, top_k_logits) / temperature
            next_token_idxs = torch.multinomial(F.softmax(softmax_logits, dim=-1), num_samples=1)
            rollouts[:, t] = next_token_idxs.squeeze(1)
            rollout_log_probs[:, t] = F.log_softmax(softmax_logits, dim=-1).gather(-1, index=next_token_idxs).squeeze(1)
            if torch.all(torch.any(rollouts == self.decoder.eos_idx, dim=-1)):
                break
        rollout_mask = self.create_inference_mask(rollouts)
        return rollouts.masked_fill(~rollout_mask, self.decoder.pad_idx), rollout_log_probs.masked_fill(~rollout_mask, 0.0), rollout_mask

    def prepare_rollouts_for_policy_theta(self, rollouts, rollout_mask):
        rollout_lens = rollout_mask.sum(dim=-1, keepdim=True)
        right_shifted_rollout_lens = rollout_lens - 1
        rollout_attention_mask = torch.arange(int(torch.max(right_shifted_rollout_lens)), device=rollouts.device).repeat([rollouts.shape[0], 1])
        rollout_attention_mask = rollout_attention_mask >= right_shifted_rollout_lens
        return rollouts[:, :-1], rollout_attention_mask

    def forward_teacher_forced(self, img_latent, latent_attention_mask, lmx_seqs, checkpoint_grads):
        input_seqs, target_seqs, lmx_attention_mask = batchify_and_split_lmx_seqs(lmx_seqs, self.decoder.pad_idx, img_latent.device)
        pred = self.decoder(input_seqs, img_latent, lmx_attention_mask, latent_attention_mask, checkpoint_grads=checkpoint_grads)
        return pred, target_seqs

    def batch_policy_inference(self, imgs, max_actions, top_k, temperature):
        img_latent, latent_attention_mask = self.encoder(imgs)
        # the reference calls self.forward_rollout_policy here, a method it does not define (M:977); the cached policy is what it means
        return self.cached_forward_rollout_policy(img_latent, latent_attention_mask, max_actions, top_k, temperature)

    def cached_forward_rollout_policy(self, img_latent, latent_attention_mask, max_actions=768, top_k=50, temperature=1.2, group_size=None,
                                      uniforms=None):
        """Sampling rollouts with KV caching (M:988-1049): per step keep the top_k logits, draw from softmax(kept / temperature), record
        log_softmax(kept)[drawn]; rows stop mattering after their first <eos>.  Returns rollouts (R, T') int64, rollout_log_probs (R, T') fp32,
        rollout_mask (R, T') bool with padding / zero log-probs outside the mask.

        The whole loop is replayed hipGraphs of `acai_decode_sample_step`.  torch.multinomial's random stream cannot be reproduced: the draws
        are inverse-CDF samples from `uniforms` (R, max_actions) in [0, 1), taken from torch's generator when None (so torch.manual_seed makes
        a rollout reproducible).  group_size (extension): img_latent rows r*group_size .. are the copies expand_img_latent_for_rollout made of
        one image; their cross K/V is then projected and stored once per image instead of once per rollout."""
        blocks = self.decoder.decoder_blocks
        if not isinstance(blocks, CachedTransformerDecoder):
            raise RuntimeError("Trying to use cached inference pathway with an uncached TransformerDecoder instance")
        G = 1 if group_size is None else int(group_size)
        if img_latent.shape[0] % G:
            raise ValueError(f"{img_latent.shape[0]} rollout rows are not a multiple of group_size {G}")
        lat = img_latent[::G] if G > 1 else img_latent
        msk = latent_attention_mask[::G] if (G > 1 and latent_attention_mask is not None) else latent_attention_mask
        mem32, lens = EG.unpad_rows(lat, msk)
        blocks.prepare_caches_packed(mem32, None, lens, group_size=G)
        eng = blocks.engine(self.decoder.pos_embedding.device)
        seqs, lps, _ = eng.sample(max_actions, top_k, temperature, uniforms=uniforms)
        return self.mask_and_clip_seqs(seqs.clone(), lps.clone())


class TeacherForcedViTOMR(ViTOMR):
    """ViTOMR assembled from a (pre-trained MAE) encoder, a transition head and an OMRDecoder (M:649-781)."""

    def __init__(self, omr_encoder, pretrained_mae_state_dict, omr_decoder, transition_head_dim=4096, transition_head_dropout=0.05):
        encoder, decoder = omr_encoder, omr_decoder
        transition_head = _TransitionHead(nn.Linear(encoder.hidden_dim, transition_head_dim), nn.GELU(), nn.Dropout(transition_head_dropout),
                                          nn.Linear(transition_head_dim, decoder.hidden_dim))
        super().__init__(encoder, transition_head, decoder)
        if pretrained_mae_state_dict:
            encoder.load_state_dict(self.create_omr_encoder_state_dict_from_mae(pretrained_mae_state_dict))
        # freezing rules (M:667-677)
        if isinstance(self.encoder, FineTuneOMREncoder) and self.encoder.frozen_blocks:
            for p in self.encoder.frozen_blocks.parameters():
                p.requires_grad = False
            for p in self.encoder.projection.parameters():
                p.requires_grad = False
            self.encoder.pos_embedding.requires_grad = False
        elif isinstance(self.encoder, OMREncoder) and not isinstance(self.encoder, FineTuneOMREncoder):
            for p in self.encoder.parameters():
                p.requires_grad = False

    def create_omr_encoder_state_dict_from_mae(self, pretrained_mae_state_dict):
        """MAE 'encoder.*' keys -> this encoder's keys; for a FineTuneOMREncoder the first num_layers - fine_tune_depth
        layers go to frozen_blocks and the rest, renumbered from 0, to fine_tune_blocks (M:679-713)."""
        sd = {k[len("encoder."):]: v for k, v in pretrained_mae_state_dict.items() if k.startswith("encoder.")}
        if not isinstance(self.encoder, FineTuneOMREncoder):
            return sd
        thr = self.encoder.num_layers - self.encoder.fine_tune_depth
        out = {}
        for k, v in sd.items():
            m = re.match(r"encoder_blocks\.layers\.(\d+)\.(.*)", k)
            if m:
                n = int(m.group(1))
                if n < thr:
                    out[f"frozen_blocks.layers.{n}.{m.group(2)}"] = v
                else:
                    out[f"fine_tune_blocks.layers.{n - thr}.{m.group(2)}"] = v
            elif k in ("encoder_blocks.norm.weight", "encoder_blocks.norm.bias"):
                out[k.replace("encoder_blocks", "fine_tune_blocks")] = v
            else:
                out[k] = v
        return out

    def forward(self, x):
        """x: list of (image, lmx_sequence) -> pred (B, L_lmxmax, V), target_seqs (B, L_lmxmax) (M:722-736)."""
        imgs, lmx_seqs = zip(*x)
        img_latent, latent_attention_mask = self.encoder(imgs)
        img_latent = self.transition_head(img_latent)
        input_seqs, target_seqs, lmx_attention_mask = batchify_and_split_lmx_seqs(lmx_seqs, self.decoder.pad_idx, img_latent.device)
        pred = self.decoder(input_seqs, img_latent, lmx_attention_mask, latent_attention_mask)
        return pred, target_seqs

    def generate(self, img_latent: torch.Tensor, seqs: torch.Tensor):
        img_latent = img_latent.expand(seqs.shape[0], -1, -1)
        logits = self.decoder.generate(seqs, img_latent)
        return F.log_softmax(logits[:, -1, :].float(), dim=-1)

    def create_fine_tune_param_groups(self, base_lr: float, fine_tune_base_lr: float, fine_tune_decay_factor: float):
        """AdamW parameter groups with layer-wise LR decay over the fine-tuned encoder blocks, last block first (M:761-781)."""
        groups = [{"params": self.decoder.parameters(), "lr": base_lr}, {"params": self.transition_head.parameters(), "lr": base_lr}]
        layer_lrs = []
        for i, layer in enumerate(reversed(self.encoder.fine_tune_blocks.layers)):
            lr = fine_tune_base_lr * (fine_tune_decay_factor ** i)
            groups.append({"params": layer.parameters(), "lr": lr})
            layer_lrs.append(lr)
        groups.append({"params": self.encoder.fine_tune_blocks.norm.parameters(), "lr": fine_tune_base_lr})
        groups.append({"params": (p for p in [self.encoder.pos_embedding]), "lr": layer_lrs[-1]})
        groups.append({"params": self.encoder.projection.parameters(), "lr": layer_lrs[-1]})
        return groups, layer_lrs


class OMRCELoss(nn.Module):
    """Cross entropy over the LMX vocabulary, <pad> targets ignored, mean over the rest (M:784-796)."""

    def __init__(self, pad_idx, label_smoothing=0.0):
        super().__init__()
        self.pad_idx = pad_idx
        self.label_smoothing = label_smoothing

    def forward(self, pred, target_seqs):
        from ..train import autograd_path
        return autograd_path.ce_loss(pred, target_seqs, self.pad_idx, self.label_smoothing)


class ScheduledSamplingViTOMR(TeacherForcedViTOMR):
    def sample_and_mix_seqs(self, teacher_forcing_prob, tf_input_seqs, tf_pred_logits, sample_tau, use_hard_sampling, device):
        """Mix gold embeddings with expected embeddings of a Gumbel-softmax sample of the first pass (M:801-817).
        Tiny (B,T,227)x(227,E) work; stays in PyTorch-ROCm as SURVEY section 2.2 allows."""
        from ..train import autograd_path as AP
        sample_mask = torch.rand(tf_input_seqs.shape, device=device) < (1 - teacher_forcing_prob)
        dec = self.decoder
        W = dec.vocab_embedding.weight
        B, T = tf_input_seqs.shape
        # nn.Embedding (M:805) as a row gather of the HIP path; the <pad> row is read through a detached copy (padding_idx: it gets no gradient)
        tok = tf_input_seqs.to(device).reshape(-1)
        table = torch.cat([W, W[dec.pad_idx:dec.pad_idx + 1].detach()], 0)
        idx = torch.where(tok == dec.pad_idx, torch.full_like(tok, W.shape[0]), tok).to(torch.int32).contiguous()
        gold = AP.GatherRowsFn.apply(table, idx, None).view(B, T, -1)
        distr = F.gumbel_softmax(tf_pred_logits.float(), tau=sample_tau, hard=use_hard_sampling)
        # (B, T, V) x (V, E) (M:809) on the path's own GEMM kernels, forward and both gradients (round 2 left it to ATen / hipBLASLt)
        expected = AP.MatmulKNFn.apply(distr.reshape(B * T, -1), W, AP._prec(), _wc(dec)).view(B, T, -1).to(gold.dtype)
        expected = torch.cat([gold[:, 0:1, :], expected], dim=1)[:, :-1]
        return torch.where(sample_mask.unsqueeze(-1), expected, gold)

    def forward_train(self, x, teacher_forcing_prob: float, sample_tau: float, use_hard_sampling: bool):
        imgs, lmx_seqs = zip(*x)
        img_latent, latent_attention_mask = self.encoder(imgs)
        img_latent = self.transition_head(img_latent)
        device = img_latent.device
        tf_input_seqs, target_seqs, lmx_attention_mask = batchify_and_split_lmx_seqs(lmx_seqs, self.decoder.pad_idx, device)
        from ..train.autograd_path import shared_cross_kv
        with shared_cross_kv():   # both passes attend to the same latent: its packed form and cross K/V projections are computed once
            tf_pred_logits = self.decoder(tf_input_seqs, img_latent, lmx_attention_mask, latent_attention_mask)
            mixed = self.sample_and_mix_seqs(teacher_forcing_prob, tf_input_seqs, tf_pred_logits, sample_tau, use_hard_sampling, device)
            pred = self.decoder(mixed, img_latent, lmx_attention_mask, latent_attention_mask, token_idxs_input=False)
        return pred, target_seqs

    def forward_eval(self, x):
        return super().forward(x)
