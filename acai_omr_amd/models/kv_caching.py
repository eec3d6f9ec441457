"""MI355X backend mirror of `acai_omr.models.kv_caching` (reference: acai_omr/models/kv_caching.py).

Same class names, constructor arguments, method names, return contracts, error behaviour and state_dict keys
(caches are non-persistent), so reference callers and the reference's own tests (tests/test_kv_caching.py) read the
same.  All arithmetic runs in the HIP library; the module-level `cached_forward` methods issue one launch per op
(API parity path), while `CachedTransformerDecoder.cached_generate` / `OMRDecoder.cached_generate` go through the
fused, hipGraph-capturable step of `engine.DecodeEngine`.  There is no CPU path: tensors must live on the GPU.

Precision follows the cache dtype, as in the reference plumbing (vitomr_inference.py:94): a bfloat16 cache means
"what autocast(bfloat16) computes" (bf16 operands/outputs, fp32 accumulate), a float cache means plain fp32.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..engine import DecodeEngine, WeightCache, unpad_rows


def _prec_of(dtype):
    if dtype == torch.bfloat16:
        return "bf16"
    if dtype in (torch.float32, torch.float):
        return "fp32"
    raise TypeError(f"cache dtype {dtype} is not supported by the MI355X backend (float32 or bfloat16)")


def _pad_pow2(dh, es):
    p = max(16 // es, 1)
    while p < dh:
        p *= 2
    return p


class KVCache(nn.Module):
    """Self-attention key/value cache, (max_batch_size, num_kv_heads, max_seq_len, head_dim) (K:5-109)."""

    def __init__(self, max_batch_size: int, max_seq_len: int, num_kv_heads: int, head_dim: int, dtype: torch.dtype) -> None:
        super().__init__()
        self._shape = (max_batch_size, num_kv_heads, max_seq_len, head_dim)
        self._dtype = dtype
        self._store = None  # allocated on first use, on the device of the first update
        self._pos = 0
        self.max_batch_size = max_batch_size

    def _alloc(self, device):
        if self._store is None or self._store[0].device != torch.device(device):
            self._store = (torch.zeros(self._shape, dtype=self._dtype, device=device), torch.zeros(self._shape, dtype=self._dtype, device=device))
        return self._store

    @property
    def k_cache(self):
        return self._alloc("cuda" if self._store is None else self._store[0].device)[0]

    @property
    def v_cache(self):
        return self._alloc("cuda" if self._store is None else self._store[0].device)[1]

    @property
    def cache_pos(self):
        return torch.arange(0, self._shape[2]) + self._pos

    def reset(self) -> None:
        if self._store is not None:
            self._store[0].zero_()
            self._store[1].zero_()
        self._pos = 0

    @property
    def size(self) -> int:
        return self._pos

    def update(self, k_val: torch.Tensor, v_val: torch.Tensor):
        cur_bsz, _, seq_len, _ = k_val.shape
        if cur_bsz > self._shape[0]:
            raise ValueError(f"The current cache has been setup with a max batch size of {self._shape[0]}"
                             f", but found new key tensors with batch size {k_val.shape[0]}!")
        assert (self._pos + seq_len) <= self._shape[2]
        k_out, v_out = self._alloc(k_val.device)
        k_out[:cur_bsz, :, self._pos:self._pos + seq_len] = k_val
        v_out[:cur_bsz, :, self._pos:self._pos + seq_len] = v_val
        self._pos += seq_len
        return k_out[:cur_bsz, :, :self._pos], v_out[:cur_bsz, :, :self._pos]


class CachedMultiheadAttention(nn.MultiheadAttention):
    """nn.MultiheadAttention as a parameter container + the single-query cached attention (K:114-140)."""

    def cached_forward(self, q_t, K_t, V_t, memory_key_padding_mask=None):
        """q_t (B,H,1,dh); K_t, V_t (B,H,T,dh); memory_key_padding_mask (B,T) True = ignore (suffix padding).
        Returns (B,1,E) = out_proj(softmax(q K^T / sqrt(dh)) V)."""
        B, H, _, dh = q_t.shape
        T = K_t.shape[2]
        prec = _prec_of(K_t.dtype)
        bf = prec == "bf16"
        dhp = _pad_pow2(dh, K_t.element_size())
        if dhp != dh:
            K_t, V_t = F.pad(K_t, (0, dhp - dh)), F.pad(V_t, (0, dhp - dh))
        K_t, V_t = K_t.contiguous(), V_t.contiguous()
        dev = q_t.device
        if memory_key_padding_mask is not None:
            lens = (~memory_key_padding_mask).sum(dim=1).to(torch.int32)
        else:
            lens = torch.full((B,), T, dtype=torch.int32, device=dev)
        # the kernel addresses head h of sequence b at off[b] + h*len[b]*dhp: with a (B,H,T,dhp) tensor and len < T the
        # head stride is T*dhp, so give every (b,h) pair its own "sequence"
        off = (torch.arange(B * H, dtype=torch.int64, device=dev) * (T * dhp))
        lens_bh = lens.to(dev).repeat_interleave(H)
        q = q_t.reshape(B * H, dh).float().contiguous()
        o = ops.decode_attn(q, K_t, V_t, off, lens_bh, 1, dh, dhp, T, round_out=bf)  # (B*H, dh)
        o = o.view(B, H * dh)
        wc = _wc(self)
        y = ops.skinny_gemm(o, wc.w(self.out_proj.weight, prec), wc.b(self.out_proj.bias, prec), round_bf16=bf)
        y = y.view(B, 1, H * dh)
        return y.to(torch.bfloat16) if bf else y


def _wc(module):
    wc = module.__dict__.get("_acai_wc")
    if wc is None:
        wc = WeightCache()
        module.__dict__["_acai_wc"] = wc
    return wc


class CachedTransformerDecoderLayer(nn.TransformerDecoderLayer):
    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1, activation=F.gelu,
                 layer_norm_eps: float = 1e-5, batch_first: bool = True, norm_first: bool = False, bias: bool = True,
                 device=None, dtype=None):
        super().__init__(d_model, nhead, dim_feedforward, dropout, activation, layer_norm_eps, batch_first, norm_first, bias, device, dtype)
        self.hidden_dim = d_model
        self.num_heads = nhead
        kw = dict(dropout=dropout, batch_first=batch_first, bias=bias, device=device, dtype=dtype)
        self.self_attn = CachedMultiheadAttention(d_model, nhead, **kw)
        self.head_dim = self.self_attn.head_dim
        self.multihead_attn = CachedMultiheadAttention(d_model, nhead, **kw)

    def cached_forward(self, tgt_t, self_attn_kv_cache: KVCache, cached_kv_mem, memory_key_padding_mask=None):
        """One decode step of one layer (K:190-223): tgt_t (B,1,E) -> (B,1,E); post-LN, no causal mask (single query)."""
        B, E, H, dh = tgt_t.shape[0], self.hidden_dim, self.num_heads, self.head_dim
        prec = _prec_of(self_attn_kv_cache._dtype)
        bf = prec == "bf16"
        wc = _wc(self)
        x = tgt_t.reshape(B, E).float().contiguous()
        sa, ca = self.self_attn, self.multihead_attn
        qkv = ops.skinny_gemm(x, wc.w(sa.in_proj_weight, prec), wc.b(sa.in_proj_bias, prec), round_bf16=bf)
        cdt = torch.bfloat16 if bf else torch.float32
        q_t, k_t, v_t = (t.reshape(B, H, 1, dh) for t in qkv.to(cdt).chunk(3, dim=-1))
        K_t, V_t = self_attn_kv_cache.update(k_t, v_t)
        sa_out = sa.cached_forward(q_t, K_t, V_t).reshape(B, E).float()
        x = ops.layernorm(x + sa_out, self.norm1.weight.detach(), self.norm1.bias.detach(), self.norm1.eps)[0]
        qc = ops.skinny_gemm(x, wc.w(ca.in_proj_weight, prec)[:E], wc.b(ca.in_proj_bias, prec)[:E], round_bf16=bf)
        K_c, V_c = cached_kv_mem
        ca_out = ca.cached_forward(qc.to(cdt).view(B, H, 1, dh), K_c, V_c, memory_key_padding_mask=memory_key_padding_mask).reshape(B, E).float()
        x = ops.layernorm(x + ca_out, self.norm2.weight.detach(), self.norm2.bias.detach(), self.norm2.eps)[0]
        h = ops.skinny_gemm(x, wc.w(self.linear1.weight, prec), wc.b(self.linear1.bias, prec), gelu=True, round_bf16=bf)
        y = ops.skinny_gemm(h, wc.w(self.linear2.weight, prec), wc.b(self.linear2.bias, prec), residual=x, round_bf16=bf)
        x = ops.layernorm(y, self.norm3.weight.detach(), self.norm3.bias.detach(), self.norm3.eps)[0]
        return x.view(B, 1, E)


class MemoryCache(nn.Module):
    """Cross-attention keys/values of the encoder memory for one layer (K:227-256)."""

    def __init__(self):
        super().__init__()
        self.K_cross = None
        self.V_cross = None

    def cache_memory_keys_and_vals(self, memory: torch.Tensor, layer: CachedTransformerDecoderLayer):
        B, S, E = memory.shape
        H, dh = layer.num_heads, layer.head_dim
        prec = "bf16" if memory.dtype == torch.bfloat16 else "fp32"
        bf = prec == "bf16"
        wc = _wc(layer)
        ca = layer.multihead_attn
        kv = ops.gemm_nt(memory.reshape(B * S, E).contiguous(), wc.w(ca.in_proj_weight, prec)[E:], wc.b(ca.in_proj_bias, prec)[E:],
                         out_dtype=memory.dtype, round_bf16=bf)
        k, v = kv[:, :E], kv[:, E:]
        self.K_cross = k.reshape(B, S, H, dh).transpose(1, 2)
        self.V_cross = v.reshape(B, S, H, dh).transpose(1, 2)

    def get_cached_keys_and_vals(self):
        return self.K_cross, self.V_cross


class CachedTransformerDecoder(nn.TransformerDecoder):
    """nn.TransformerDecoder as the parameter container + the cached inference loop (K:258-302).
    prepare_caches(memory) then cached_generate(embedding_t, mask) once per token, starting with <bos>."""

    def __init__(self, decoder_layer: CachedTransformerDecoderLayer, num_layers: int, max_batch_size: int, max_decoder_seq_len: int,
                 cache_dtype, norm: nn.Module = None):
        assert isinstance(decoder_layer, CachedTransformerDecoderLayer), "Can't use uncached TransformerDecoderLayer in a cached TransformerDecoder"
        super().__init__(decoder_layer, num_layers, norm)
        self.self_attn_caches = nn.ModuleList([KVCache(max_batch_size, max_decoder_seq_len, layer.num_heads, layer.head_dim, dtype=cache_dtype)
                                               for layer in self.layers])
        self.cross_attn_caches = nn.ModuleList([MemoryCache() for _ in self.layers])
        self.max_batch_size = max_batch_size
        self.max_decoder_seq_len = max_decoder_seq_len
        self.cache_dtype = cache_dtype
        self.__dict__["_engine"] = None
        self.__dict__["_pending"] = None
        self.__dict__["_omr"] = None  # set by OMRDecoder so that the fused step can embed / unembed

    def __getstate__(self):
        """copy.deepcopy(model) / torch.save(model) (the reference's GRPO loop deep-copies its policy, omr_grpo_train.py): the decode engine
        (ctypes descriptors with raw pointers, a stream, captured graphs) is per-instance runtime state and is rebuilt on first use."""
        st = self.__dict__.copy()
        st["_engine"] = None
        st["_pending"] = None
        st.pop("_mask_lens", None)
        return st

    def engine(self, device):
        eng = self.__dict__["_engine"]
        if eng is None or eng.device != torch.device(device):
            eng = DecodeEngine(self, self.__dict__["_omr"], self.max_batch_size, self.max_decoder_seq_len, _prec_of(self.cache_dtype), torch.device(device))
            self.__dict__["_engine"] = eng
        return eng

    def prepare_caches(self, encoder_memory):
        """Reset the self-attention caches and (lazily, once the padding mask of the first cached_generate call is known)
        compute every layer's cross K/V from `encoder_memory` (B,S,E).  K/V of padded memory rows are never computed."""
        if not encoder_memory.is_cuda:
            raise RuntimeError("acai_omr_amd: encoder_memory must be a GPU tensor (no CPU fallback)")
        if encoder_memory.shape[0] > self.max_batch_size:
            raise ValueError(f"The current cache has been setup with a max batch size of {self.max_batch_size}"
                             f", but found new key tensors with batch size {encoder_memory.shape[0]}!")
        for c in self.self_attn_caches:
            c._pos = 0
        self.__dict__["_pending"] = encoder_memory
        self.engine(encoder_memory.device).reset_self_cache()

    def prepare_caches_packed(self, mem32, memb, lens, group_size=1):
        """Packed fast path used by the inference entry points: memory is a ragged token stream.  group_size > 1: each memory serves
        that many consecutive decode rows (GRPO rollouts of one image) with ONE copy of its cross K/V."""
        self.__dict__["_pending"] = None
        self.engine(mem32.device if mem32 is not None else memb.device).prepare(mem32, memb, lens, group_size=group_size)

    def _materialise(self, memory_key_padding_mask):
        mem = self.__dict__["_pending"]
        if mem is None:
            return
        packed, lens = unpad_rows(mem, memory_key_padding_mask)
        eng = self.engine(mem.device)
        eng.prepare(packed, None, lens)
        self.__dict__["_pending"] = None
        self.__dict__["_mask_lens"] = lens

    def cached_generate(self, embedding_t, memory_key_padding_mask=None):
        self._materialise(memory_key_padding_mask)
        eng = self.engine(embedding_t.device)
        x = eng.hidden_step(embedding_t)
        for c in self.self_attn_caches:
            c._pos += 1
        x = x.view(-1, 1, eng.E).clone()
        return x
