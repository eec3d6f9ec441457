"""CPU oracle for the ViTOMR / MAE hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch CPU restatement of the algorithm the reference
(jsnchon/acai-omr) runs through `torch.nn.Transformer*` modules.  It exists so
that the HIP path can be checked on a GPU box where `/root/reference` does not
exist.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it; the product package (`acai_omr_amd`) never
does and fails loudly when its HIP extension is missing.

Parity is PINNED: `oracle/gen_golden.py` imports the real reference classes
from `/root/reference`, runs them on seeded inputs and commits inputs, weights
and outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks every
function here against those vectors (and against the reference's own KATs,
SURVEY.md section 8c).

Formulation: everything works on a PACKED token stream (sum of per-image
lengths, no padding) described by per-sequence lengths; padding only appears at
the API edge (`pad_packed`).  That is mathematically identical to the
reference's "padded batch + key padding mask" for every real token.

Precision modes (`prec`):
  "fp32" - no rounding anywhere (reference outside autocast).
  "bf16" - restates `torch.autocast("cpu", bfloat16)` as the reference's
           inference/train plumbing uses it (vitomr_inference.py:82,
           omr_teacher_force_train.py:116): `linear` and SDPA inputs/outputs
           are rounded to bf16, accumulation is fp32, the residual stream and
           LayerNorm stay fp32 (fp32 + bf16 promotes to fp32; layer_norm is not
           a CPU-autocast op), GELU of a bf16 tensor is bf16, the KV cache is
           bf16, logits are bf16.

Reference citations are `acai_omr/models/models.py` (M:) and
`acai_omr/models/kv_caching.py` (K:) line numbers.
"""
import math

import torch
import torch.nn.functional as F

NEG_INF = float("-inf")


# ----------------------------------------------------------------------------
# precision helpers
# ----------------------------------------------------------------------------
def rbf16(x):
    """Round-to-nearest-even to bf16, kept in an fp32 container."""
    return x.to(torch.bfloat16).to(torch.float32)


def _r(x, prec):
    return rbf16(x) if prec == "bf16" else x


WEIGHTS_PREROUNDED = False  # bench.py's cpu_baseline rounds the weights once (autocast caches its weight casts too)


def linear(x, w, b, prec):
    """F.linear; under prec == "bf16" the autocast cast policy (inputs, weight and
    bias to bf16, fp32 accumulate, bf16 output)."""
    if prec == "bf16":
        if WEIGHTS_PREROUNDED:
            y = rbf16(x) @ w.t()
            return rbf16(y if b is None else y + b)
        y = rbf16(x) @ rbf16(w).t()
        if b is not None:
            y = y + rbf16(b)
        return rbf16(y)
    y = x @ w.t()
    if b is not None:
        y = y + b
    return y


def gelu(x, prec):
    """Exact erf GELU (M:31 activation="gelu"); bf16 in -> bf16 out under autocast."""
    return _r(F.gelu(x), prec)


def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim, biased variance, fp32 in every mode."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def sdpa(q, k, v, keep_mask, prec):
    """softmax(q k^T / sqrt(d)) v for ONE head. q (Lq,d) k,v (Lk,d); keep_mask bool
    (Lq,Lk) True = attend, or None."""
    d = q.shape[-1]
    s = (q @ k.t()) * (1.0 / math.sqrt(d))
    if keep_mask is not None:
        s = s.masked_fill(~keep_mask, NEG_INF)
    p = torch.softmax(s, dim=-1)
    return _r(p @ v, prec)


def mha_packed(xq, xkv, lens_q, lens_k, in_w, in_b, out_w, out_b, num_heads, causal, prec,
               key_keep=None):
    """nn.MultiheadAttention on packed streams: xq (sum Lq, E), xkv (sum Lk, E).
    Each sequence attends only inside itself (== key padding mask on a padded batch).
    key_keep: optional list of bool (Lk_i,) per sequence, True = key may be attended
    (tgt_key_padding_mask restated)."""
    E = xq.shape[-1]
    dh = E // num_heads
    q = linear(xq, in_w[:E], in_b[:E], prec)
    kv = linear(xkv, in_w[E:], in_b[E:], prec)
    k, v = kv[:, :E], kv[:, E:]
    out = torch.empty_like(q)
    oq = ok = 0
    for i, (lq, lk) in enumerate(zip(lens_q, lens_k)):
        keep = None
        if causal:
            keep = torch.ones(lq, lk, dtype=torch.bool).tril()
        if key_keep is not None:
            kk = key_keep[i].unsqueeze(0).expand(lq, lk)
            keep = kk if keep is None else (keep & kk)
        for h in range(num_heads):
            sl = slice(h * dh, (h + 1) * dh)
            out[oq:oq + lq, sl] = sdpa(q[oq:oq + lq, sl], k[ok:ok + lk, sl], v[ok:ok + lk, sl], keep, prec)
        oq += lq
        ok += lk
    return linear(out, out_w, out_b, prec)


# ----------------------------------------------------------------------------
# packing helpers (API edge)
# ----------------------------------------------------------------------------
def patchify(img, P):
    """nn.Unfold(P, stride P) on a (1,H,W) image, transposed to (N, P*P) (M:48-52).
    Row i*w_p + j is patch (i, j); inside a patch pixels are row-major (kh, kw)."""
    C, H, W = img.shape
    assert C == 1
    hp, wp = H // P, W // P
    x = img[0, :hp * P, :wp * P].reshape(hp, P, wp, P).permute(0, 2, 1, 3)
    return x.reshape(hp * wp, P * P), hp, wp


def pad_packed(x, lens, pad_value=0.0):
    """packed (sum L, ...) -> (B, Lmax, ...) and the bool mask True = padding (M:70-73)."""
    B, Lm = len(lens), max(lens)
    out = x.new_full((B, Lm) + tuple(x.shape[1:]), pad_value)
    mask = torch.ones(B, Lm, dtype=torch.bool)
    o = 0
    for i, l in enumerate(lens):
        out[i, :l] = x[o:o + l]
        mask[i, :l] = False
        o += l
    return out, mask


def unpad(x, mask):
    """(B, Lmax, ...) + mask True = padding -> packed, lens."""
    lens = (~mask).sum(dim=1).tolist()
    return torch.cat([x[i, :l] for i, l in enumerate(lens)], dim=0), lens


def interpolate_pe(pe, hp, wp):
    """OMREncoder.interpolate_pe (M:291-302): bilinear, align_corners=False."""
    g = pe.permute(2, 0, 1).unsqueeze(0)
    g = F.interpolate(g, size=(hp, wp), mode="bilinear", align_corners=False)
    return g.squeeze(0).permute(1, 2, 0)


def pe_slice(pe, hp, wp, allow_interp):
    """pos_embedding[:hp,:wp].reshape(-1,E) (M:50) or the interpolated grid (M:315-318)."""
    Hm, Wm, E = pe.shape
    if hp > Hm or wp > Wm:
        if not allow_interp:
            raise ValueError(f"{hp} x {wp} image is too large for max positional embedding grid of shape {Hm} x {Wm}")
        return interpolate_pe(pe, hp, wp).reshape(-1, E)
    return pe[:hp, :wp, :].reshape(-1, E)


# ----------------------------------------------------------------------------
# transformer stacks (post-LN, M:30-34 / torch transformer.py:952-956)
# ----------------------------------------------------------------------------
def encoder_layer(x, lens, sd, p, num_heads, prec):
    a = mha_packed(x, x, lens, lens, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                   sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], num_heads, False, prec)
    x = layer_norm(x + a, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    h = gelu(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"], prec), prec)
    h = linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"], prec)
    return layer_norm(x + h, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)


def encoder_stack(x, lens, sd, prefix, num_heads, prec):
    """nn.TransformerEncoder: layers.{i}.* then optional final norm (eps 1e-6, M:33)."""
    i = 0
    while f"{prefix}layers.{i}.norm1.weight" in sd:
        x = encoder_layer(x, lens, sd, f"{prefix}layers.{i}.", num_heads, prec)
        i += 1
    if prefix + "norm.weight" in sd:
        x = layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6)
    return x


def decoder_layer_tf(x, mem, lens_t, lens_s, sd, p, num_heads, prec, tgt_keep):
    """nn.TransformerDecoderLayer teacher-forced (torch transformer.py:1144-1153)."""
    a = mha_packed(x, x, lens_t, lens_t, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                   sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], num_heads, True, prec,
                   key_keep=tgt_keep)
    x = layer_norm(x + a, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    c = mha_packed(x, mem, lens_t, lens_s, sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"],
                   sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"], num_heads, False, prec)
    x = layer_norm(x + c, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = gelu(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"], prec), prec)
    h = linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"], prec)
    return layer_norm(x + h, sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)


# ----------------------------------------------------------------------------
# A1/A2: Encoder / OMREncoder / FineTuneOMREncoder (M:14-96, 290-376)
# ----------------------------------------------------------------------------
def encoder_embed(imgs, sd, prefix, P, allow_interp, prec):
    """batchify (M:36-66 / M:304-332) on the packed stream: projection + PE."""
    rows, pes, lens = [], [], []
    pe = sd[prefix + "pos_embedding"]
    for img in imgs:
        x, hp, wp = patchify(img, P)
        if not allow_interp and (hp > pe.shape[0] or wp > pe.shape[1]):
            raise ValueError(f"{hp} x {wp} image is too large for max positional embedding grid of shape {pe.shape[0]} x {pe.shape[1]}")
        rows.append(x)
        pes.append(pe_slice(pe, hp, wp, allow_interp))
        lens.append(hp * wp)
    x = torch.cat(rows, 0)
    x = linear(x, sd[prefix + "projection.weight"], sd[prefix + "projection.bias"], prec) + torch.cat(pes, 0)
    return x, lens


def encoder_forward(imgs, sd, prefix, P, num_heads, kind="omr_ft", prec="fp32"):
    """kind: "base" (Encoder, encoder_blocks), "omr" (OMREncoder, interpolation allowed),
    "omr_ft" (FineTuneOMREncoder: frozen_blocks then fine_tune_blocks).  Returns packed latent, lens."""
    x, lens = encoder_embed(imgs, sd, prefix, P, allow_interp=(kind != "base"), prec=prec)
    if kind == "omr_ft":
        if f"{prefix}frozen_blocks.layers.0.norm1.weight" in sd:
            x = encoder_stack(x, lens, sd, prefix + "frozen_blocks.", num_heads, prec)
        x = encoder_stack(x, lens, sd, prefix + "fine_tune_blocks.", num_heads, prec)
    else:
        x = encoder_stack(x, lens, sd, prefix + "encoder_blocks.", num_heads, prec)
    return x, lens


def encoder_forward_padded(imgs, sd, prefix, P, num_heads, kind="omr_ft", prec="fp32", final_norm_bias_fill=None):
    """API-edge view: (B, Lmax, E) + mask.  In eval/no-grad torch takes the nested-tensor
    fast path (transformer.py:529-550): padded rows are zero before the final LayerNorm,
    i.e. they come out as that norm's bias; pass final_norm_bias_fill=<bias> to restate it."""
    x, lens = encoder_forward(imgs, sd, prefix, P, num_heads, kind, prec)
    out, mask = pad_packed(x, lens)
    if final_norm_bias_fill is not None:
        out[mask] = final_norm_bias_fill
    return out, mask


def transition_head(x, sd, prec, prefix="transition_head."):
    """Linear, GELU, Dropout(eval: identity), Linear (M:655-660)."""
    h = gelu(linear(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"], prec), prec)
    return linear(h, sd[prefix + "3.weight"], sd[prefix + "3.bias"], prec)


# ----------------------------------------------------------------------------
# A6/A7: OMRDecoder teacher-forced + KV-cached decode (M:378-528, K:5-302)
# ----------------------------------------------------------------------------
def decoder_forward_tf(inputs, mem, lens_t, lens_s, sd, num_heads, prec, prefix="decoder.",
                       token_idxs_input=True, tgt_pad_keep=None):
    """OMRDecoder.forward on packed streams (M:445-483).  inputs: packed token ids (sum T,)
    or packed embeddings (sum T, E).  Positions restart at 0 for each sequence (M:465-466).
    tgt_pad_keep: list of bool (T_i,) True = not <pad> (tgt_key_padding_mask inverted)."""
    if token_idxs_input:
        x = sd[prefix + "vocab_embedding.weight"][inputs]
    else:
        x = inputs
    pos = torch.cat([torch.arange(t) for t in lens_t])
    x = x + sd[prefix + "pos_embedding"][pos]
    i = 0
    while f"{prefix}decoder_blocks.layers.{i}.norm1.weight" in sd:
        x = decoder_layer_tf(x, mem, lens_t, lens_s, sd, f"{prefix}decoder_blocks.layers.{i}.", num_heads, prec, tgt_pad_keep)
        i += 1
    x = layer_norm(x, sd[prefix + "decoder_blocks.norm.weight"], sd[prefix + "decoder_blocks.norm.bias"], 1e-6)
    return linear(x, sd[prefix + "unembed.weight"], sd[prefix + "unembed.bias"], prec)


class DecodeState:
    """Self K/V caches (K:5-109) and cross K/V (K:227-256), one entry per layer.  Heads are batched into one matmul
    per sequence so that the CPU baseline is not a Python-loop artefact."""

    def __init__(self, mem, lens_s, sd, num_heads, prec, prefix="decoder.", t_cap=64):
        self.sd, self.num_heads, self.prec, self.prefix = sd, num_heads, prec, prefix
        self.lens_s = lens_s
        self.B = len(lens_s)
        self.L = 0
        while f"{prefix}decoder_blocks.layers.{self.L}.norm1.weight" in sd:
            self.L += 1
        E = sd[prefix + "pos_embedding"].shape[1]
        self.E = E
        H, dh = num_heads, E // num_heads
        # MemoryCache.cache_memory_keys_and_vals (K:235-253): rows d..3d of the cross in_proj; stored (H, S_b, dh) per sequence
        self.k_cross, self.v_cross = [], []
        for l in range(self.L):
            p = f"{prefix}decoder_blocks.layers.{l}.multihead_attn."
            kv = linear(mem, sd[p + "in_proj_weight"][E:], sd[p + "in_proj_bias"][E:], prec)
            ks, vs, o = [], [], 0
            for ls in lens_s:
                ks.append(kv[o:o + ls, :E].reshape(ls, H, dh).transpose(0, 1).contiguous())
                vs.append(kv[o:o + ls, E:].reshape(ls, H, dh).transpose(0, 1).contiguous())
                o += ls
            self.k_cross.append(ks)
            self.v_cross.append(vs)
        self.t = 0
        self.k_self = [torch.zeros(self.B, H, t_cap, dh) for _ in range(self.L)]
        self.v_self = [torch.zeros(self.B, H, t_cap, dh) for _ in range(self.L)]

    def _grow(self):
        for c in (self.k_self, self.v_self):
            for l in range(self.L):
                c[l] = torch.cat([c[l], torch.zeros_like(c[l])], dim=2)


def _sdpa_heads(q, K, V, prec):
    """q (..., H, 1, dh), K/V (..., H, T, dh): softmax(q K^T / sqrt(dh)) V for every head at once, no mask (K:206)."""
    s = (q @ K.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    return _r(torch.softmax(s, dim=-1) @ V, prec)


def decode_step(state, tokens, time_step):
    """OMRDecoder.cached_generate (M:518-528) + CachedTransformerDecoder.cached_generate
    (K:292-302) + layer.cached_forward (K:190-223).  tokens (B,), time_step indexes
    pos_embedding literally.  Returns logits (B, V)."""
    sd, prec, H, E, px, B = state.sd, state.prec, state.num_heads, state.E, state.prefix, state.B
    dh = E // H
    if state.t >= state.k_self[0].shape[2]:
        state._grow()
    t = state.t
    x = sd[px + "vocab_embedding.weight"][tokens] + sd[px + "pos_embedding"][time_step]
    for l in range(state.L):
        p = f"{px}decoder_blocks.layers.{l}."
        qkv = linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"], prec)
        q, k, v = (c.reshape(B, H, 1, dh) for c in (qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]))
        state.k_self[l][:, :, t:t + 1] = k
        state.v_self[l][:, :, t:t + 1] = v
        sa = _sdpa_heads(q, state.k_self[l][:, :, :t + 1], state.v_self[l][:, :, :t + 1], prec).reshape(B, E)
        sa = linear(sa, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], prec)
        x = layer_norm(x + sa, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        qc = linear(x, sd[p + "multihead_attn.in_proj_weight"][:E], sd[p + "multihead_attn.in_proj_bias"][:E], prec)
        ca = torch.stack([_sdpa_heads(qc[b].reshape(H, 1, dh), state.k_cross[l][b], state.v_cross[l][b], prec).reshape(E)
                          for b in range(B)])
        ca = linear(ca, sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"], prec)
        x = layer_norm(x + ca, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        h1 = gelu(linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"], prec), prec)
        h2 = linear(h1, sd[p + "linear2.weight"], sd[p + "linear2.bias"], prec)
        x = layer_norm(x + h2, sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    state.t += 1
    x = layer_norm(x, sd[px + "decoder_blocks.norm.weight"], sd[px + "decoder_blocks.norm.bias"], 1e-6)
    return linear(x, sd[px + "unembed.weight"], sd[px + "unembed.bias"], prec)


def next_token(logits, prec):
    """cached_get_next_token (M:579-581): argmax (first index on ties), log_softmax, gather.
    Under autocast the logits are bf16 and so is log_softmax's output."""
    idx = torch.argmax(logits, dim=-1)
    lp = _r(F.log_softmax(logits, dim=-1), prec)
    return idx, lp.gather(-1, idx.unsqueeze(1)).squeeze(1)


def inference_mask(seqs, eos_idx):
    """ViTOMR.create_inference_mask (M:550-559)."""
    eos = seqs == eos_idx
    seen = eos.int().cumsum(dim=-1)
    return (seen == 0) | (eos & (seen == 1))


def mask_and_clip(seqs, lps, eos_idx, pad_idx):
    """ViTOMR.mask_and_clip_seqs (M:585-596)."""
    m = inference_mask(seqs, eos_idx)
    seqs = seqs.masked_fill(~m, pad_idx)
    lps = lps.masked_fill(~m, 0.0)
    n = int(m.sum(dim=-1).max())
    return seqs[:, :n], lps[:, :n], m[:, :n]


def greedy_generate(mem, lens_s, sd, num_heads, prec, max_len, bos_idx=0, pad_idx=1, eos_idx=2,
                    prefix="decoder.", return_logits=False):
    """ViTOMR.cached_greedy_generate (M:600-615), including quirk Q1: the token at index
    t-1 is embedded with pos_embedding[t] (M:576)."""
    B = len(lens_s)
    state = DecodeState(mem, lens_s, sd, num_heads, prec, prefix)
    seqs = torch.full((B, max_len), pad_idx, dtype=torch.long)
    seqs[:, 0] = bos_idx
    lps = torch.zeros(B, max_len)
    finished = torch.zeros(B, dtype=torch.bool)
    all_logits = []
    for t in range(1, max_len):
        logits = decode_step(state, seqs[:, t - 1], t)
        if return_logits:
            all_logits.append(logits)
        idx, lp = next_token(logits, prec)
        seqs[:, t] = idx
        lps[:, t] = lp
        finished |= idx == eos_idx
        if bool(finished.all()):
            break
    out = mask_and_clip(seqs, lps, eos_idx, pad_idx)
    if return_logits:
        return out + (torch.stack(all_logits, 1),)
    return out


def vitomr_inference(imgs, sd, enc_heads, dec_heads, P, max_len, prec_decoder="bf16", enc_kind="omr_ft"):
    """vitomr_inference.inference (acai_omr/inference/vitomr_inference.py:73-86): encoder in
    fp32 outside autocast, transition head + greedy decode under autocast(bf16)."""
    lat, lens = encoder_forward(imgs, sd, "encoder.", P, enc_heads, enc_kind, "fp32")
    mem = transition_head(lat, sd, prec_decoder)
    return greedy_generate(mem, lens, sd, dec_heads, prec_decoder, max_len)


# ----------------------------------------------------------------------------
# A9: teacher forcing, CE loss (M:531-540, 649-838)
# ----------------------------------------------------------------------------
def batchify_and_split_lmx_seqs(lmx_seqs, pad_idx):
    """M:531-540 on plain tensors."""
    B, Lm = len(lmx_seqs), max(len(s) for s in lmx_seqs)
    full = torch.full((B, Lm), pad_idx, dtype=lmx_seqs[0].dtype)
    for i, s in enumerate(lmx_seqs):
        full[i, :len(s)] = s
    inp, tgt = full[:, :-1], full[:, 1:]
    return inp, tgt, inp == pad_idx


def teacher_forced_forward(batch, sd, enc_heads, dec_heads, P, prec, enc_kind="omr_ft", pad_idx=1):
    """TeacherForcedViTOMR.forward (M:722-736).  Returns padded pred (B, T, V), target (B, T)."""
    imgs, lmx = zip(*batch)
    lat, lens_s = encoder_forward(list(imgs), sd, "encoder.", P, enc_heads, enc_kind, prec)
    mem = transition_head(lat, sd, prec)
    inp, tgt, pad_mask = batchify_and_split_lmx_seqs(lmx, pad_idx)
    B, T = inp.shape
    # the reference runs every row at the padded length T with <pad> keys masked
    lens_t = [T] * B
    keep = [~pad_mask[i] for i in range(B)]
    pred = decoder_forward_tf(inp.reshape(-1), mem, lens_t, lens_s, sd, dec_heads, prec, tgt_pad_keep=keep)
    return pred.reshape(B, T, -1), tgt


def ce_loss(pred, target, pad_idx=1, label_smoothing=0.0):
    """OMRCELoss (M:784-796): mean over non-pad targets of (1 - eps) * nll + eps / V * sum_c(-log p_c) (nn.CrossEntropyLoss with
    ignore_index and label_smoothing = eps, restated; eps = 0 is the plain NLL)."""
    lp = torch.log_softmax(pred.reshape(-1, pred.shape[-1]).float(), dim=-1)
    tg = target.reshape(-1)
    keep = tg != pad_idx
    nll = -lp.gather(-1, tg.clamp(min=0).unsqueeze(1)).squeeze(1)
    smooth = -lp.mean(dim=-1)
    per = (1.0 - label_smoothing) * nll + label_smoothing * smooth
    return (per * keep).sum() / keep.sum()


# ----------------------------------------------------------------------------
# A3/A4/A5: MAE (M:100-288)
# ----------------------------------------------------------------------------
def mae_mask_ids(noise, mask_ratio):
    """mask_sequence (M:106-119) from injected noise: ids_keep, ids_restore, seq_mask (int32, 1 = masked)."""
    N = noise.shape[0]
    keep = int(N * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise)
    ids_restore = torch.argsort(ids_shuffle)
    seq_mask = torch.ones(N, dtype=torch.int32)
    seq_mask[:keep] = 0
    return ids_shuffle[:keep], ids_restore, seq_mask[ids_restore], keep


def mae_forward(batch, noises, sd, P, mask_ratio, enc_heads, dec_heads, prec="fp32"):
    """MAE.forward (M:249-269) with injected per-image noise.  Returns packed
    pred (sum N, P*P), packed loss_mask (sum N,) bool, packed target (sum N, P*P), lens."""
    xs, ys = zip(*batch)
    pe = sd["encoder.pos_embedding"]
    rows, pes, kept, lens, restores, masks, dpes = [], [], [], [], [], [], []
    dpe = sd["decoder_pos_embedding"]
    for img, noise in zip(xs, noises):
        x, hp, wp = patchify(img, P)
        if hp > pe.shape[0] or wp > pe.shape[1]:
            raise ValueError(f"{hp} x {wp} image is too large for max positional embedding grid of shape {pe.shape[0]} x {pe.shape[1]}")
        ids_keep, ids_restore, seq_mask, k = mae_mask_ids(noise, mask_ratio)
        rows.append(x[ids_keep])
        pes.append(pe_slice(pe, hp, wp, False)[ids_keep])
        kept.append(k)
        lens.append(hp * wp)
        restores.append(ids_restore)
        masks.append(seq_mask.bool())
        dpes.append(pe_slice(dpe, hp, wp, False))
    x = linear(torch.cat(rows, 0), sd["encoder.projection.weight"], sd["encoder.projection.bias"], prec) + torch.cat(pes, 0)
    x = encoder_stack(x, kept, sd, "encoder.encoder_blocks.", enc_heads, prec)
    x = linear(x, sd["decoder_embed.weight"], sd["decoder_embed.bias"], prec)
    # prepare_for_decoder (M:219-241): append mask tokens, unshuffle, add decoder PE
    seqs, o = [], 0
    mt = sd["mask_token"].reshape(1, -1)
    for k, n, r, d in zip(kept, lens, restores, dpes):
        s = torch.cat([x[o:o + k], mt.expand(n - k, -1)], 0)[r] + d
        seqs.append(s)
        o += k
    x = encoder_stack(torch.cat(seqs, 0), lens, sd, "decoder.decoder_blocks.", dec_heads, prec)
    pred = linear(x, sd["decoder_unembed.weight"], sd["decoder_unembed.bias"], prec)
    target = torch.cat([patchify(t, P)[0] for t in ys], 0)
    return pred, torch.cat(masks, 0), target, lens


def mae_loss(pred, loss_mask, target):
    """MAELoss (M:273-288): unbiased variance, eps inside the sqrt, masked mean."""
    mean = target.mean(dim=-1, keepdim=True)
    var = target.var(dim=-1, keepdim=True)
    t = (target - mean) / (var + 1.0e-6) ** 0.5
    loss = ((pred - t) ** 2).mean(dim=-1)
    lm = loss_mask.to(loss.dtype)
    return (loss * lm).sum() / lm.sum()


# ---- GRPO rollout policy (reference acai_omr/models/models.py:988-1049) ------------------------------------------------------------------
def rollout_sample_step(logits, u, top_k, temperature, round_lp=False):
    """One sampling step on (R, V) logits with uniforms u (R,): top-k filter (M:1003), softmax(kept / temperature) (M:1006-1007), draw,
    log_softmax(kept)[drawn] (M:1017-1018).  The reference draws with torch.multinomial; this restatement - and the HIP kernel - draw by
    inverse CDF over the kept logits in descending order (ties: lower index first), which has the same distribution."""
    logits = logits.float()
    vals, idx = torch.sort(logits, dim=-1, descending=True, stable=True)
    vals, idx = vals[:, :top_k], idx[:, :top_k]
    m = vals[:, :1]
    pT = torch.exp((vals - m) / temperature)
    cdf = torch.cumsum(pT, dim=-1)
    target = u.float().unsqueeze(1) * pT.sum(-1, keepdim=True)
    hit = cdf > target
    r = torch.where(hit.any(-1), hit.float().argmax(-1), torch.full((logits.shape[0],), vals.shape[1] - 1))
    tok = idx.gather(-1, r.unsqueeze(1)).squeeze(1)
    lp = (vals.gather(-1, r.unsqueeze(1)).squeeze(1) - m.squeeze(1)) - torch.log(torch.exp(vals - m).sum(-1))
    if round_lp:
        lp = lp.to(torch.bfloat16).float()
    return tok, lp


def rollout_uniform_for_token(logits, tok, top_k, temperature):
    """The uniform u for which `rollout_sample_step(logits, u, ...)` draws `tok` (R,): the midpoint of tok's interval of the tempered CDF
    over the kept logits in descending order (ties: lower index first).  Lets a rollout the REFERENCE drew with torch.multinomial
    (M:1008) be replayed through the inverse-CDF draw of this restatement and of the HIP kernel.  Also returns whether tok was kept."""
    logits = logits.float()
    vals, idx = torch.sort(logits, dim=-1, descending=True, stable=True)
    vals, idx = vals[:, :top_k], idx[:, :top_k]
    pT = torch.exp((vals - vals[:, :1]) / temperature)
    cdf = torch.cumsum(pT, dim=-1)
    hit = idx == tok.unsqueeze(1)
    kept = hit.any(-1)
    r = hit.float().argmax(-1)
    hi = cdf.gather(-1, r.unsqueeze(1)).squeeze(1)
    lo = hi - pT.gather(-1, r.unsqueeze(1)).squeeze(1)
    return (0.5 * (lo + hi) / pT.sum(-1)).clamp(0.0, 1.0 - 1e-7), kept


def rollout_generate(mem, lens_s, sd, num_heads, prec, max_actions, top_k, temperature, uniforms, bos_idx=0, pad_idx=1, eos_idx=2,
                     prefix="decoder.", return_logits=False):
    """GRPOViTOMR.cached_forward_rollout_policy (M:988-1049) with the draw of every step taken from `uniforms` (R, max_actions): the greedy
    loop's plumbing (quirk Q1 included, M:1000) with `rollout_sample_step` in place of the arg-max; rows after their first <eos> are
    padded / zeroed and the result is clipped to the longest rollout (M:1036-1047)."""
    R = len(lens_s)
    state = DecodeState(mem, lens_s, sd, num_heads, prec, prefix)
    ro = torch.full((R, max_actions), pad_idx, dtype=torch.long)
    ro[:, 0] = bos_idx
    lps = torch.zeros(R, max_actions)
    finished = torch.zeros(R, dtype=torch.bool)
    all_logits = []
    for t in range(1, max_actions):
        logits = decode_step(state, ro[:, t - 1], t)
        if return_logits:
            all_logits.append(logits)
        tok, lp = rollout_sample_step(logits, uniforms[:, t], top_k, temperature, round_lp=(prec == "bf16"))
        ro[:, t] = tok
        lps[:, t] = lp
        finished |= tok == eos_idx
        if bool(finished.all()):
            break
    out = mask_and_clip(ro, lps, eos_idx, pad_idx)
    if return_logits:
        return out + (torch.stack(all_logits, 1),)
    return out
