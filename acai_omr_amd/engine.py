"""Packed-stream execution of the hot path on the HIP library (host orchestration only).

The mirror modules in `acai_omr_amd.models` keep the reference's nn.Module surface and state_dict keys; their
parameters live in stock torch containers (nn.TransformerEncoder/Decoder are used as PARAMETER CONTAINERS only,
their forward is never called).  This file turns those parameters into kernel launches:

  encoder_stack   post-LN ViT blocks on a packed token stream (cu_seqlens, no padding)   [models.py:30-34,76-79]
  DecodeEngine    cross-K/V prefill + hipGraph-replayed greedy decode                      [kv_caching.py:190-302, models.py:562-615]

Precision ("fp32" | "bf16") follows the reference plumbing: fp32 = outside autocast; bf16 = what
torch.autocast(bfloat16) does to linear / SDPA (bf16 operands and outputs, fp32 accumulate, fp32 residual + LayerNorm).
"""
import ctypes
import os

import weakref

import torch

from . import _lib, ops


class WeightCache:
    """bf16 operand copies of fp32 master parameters (autocast's weight cast, done once instead of per call) and
    bf16-rounded fp32 biases.  Entries are refreshed when the parameter's version counter or storage changes."""

    def __init__(self):
        self._c = {}

    # derived data keyed on parameter identity: a copied / pickled module starts with an empty cache
    def __deepcopy__(self, memo):
        return WeightCache()

    def __reduce__(self):
        return (WeightCache, ())

    def invalidate(self):
        """Drop every cached copy.  Needed after writes that do not move a parameter's version counter (`p.data.copy_`, EMA through `.data`,
        weight surgery): the cache keys on (version, storage pointer).  `load_state_dict` and optimizer steps bump versions and need no call."""
        self._c.clear()

    def _get(self, p, kind, make):
        key = (id(p), kind)
        tag = (p._version, p.data_ptr())
        hit = self._c.get(key)
        if hit is not None and hit[0] != tag and kind in self._BATCHED and p.is_cuda and not torch.cuda.is_current_stream_capturing():
            self._refresh_stale(p.device)      # an optimizer step went by: every stale copy on this device in ONE launch
            hit = self._c.get(key)
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                hit = (tag, make(p.detach()), weakref.ref(p))
            self._c[key] = hit
        return hit[1]

    _BATCHED = ("w16", "wtbf16", "b16")

    def _refresh_stale(self, device):
        """Rebuild every cached bf16 / transposed-bf16 / rounded-bias copy whose parameter moved on (version or storage) with one
        `acai_cast_weights` launch (through ATen: one cast or copy launch per tensor and kind, ~300 per training step).  Fresh output tensors:
        a copy still referenced by an autograd graph keeps its old values."""
        groups = {}
        for (pid, kind), (tag, _, ref) in list(self._c.items()):
            p = ref()
            if p is None:
                del self._c[(pid, kind)]
                continue
            if kind not in self._BATCHED or p.device != device or tag == (p._version, p.data_ptr()):
                continue
            if p.dtype != torch.float32 or not p.is_contiguous() or p.dim() != (1 if kind == "b16" else 2):
                continue
            groups.setdefault(pid, (p, set()))[1].add(kind)
        if not groups:
            return
        order = list(groups.values())
        with torch.no_grad():
            outs = ops.cast_weights([(p.detach(), "w16" in ks, "wtbf16" in ks, "b16" in ks) for p, ks in order])
        for (p, ks), (d16, d16t, d32) in zip(order, outs):
            tag = (p._version, p.data_ptr())
            for kind, t in (("w16", d16), ("wtbf16", d16t), ("b16", d32)):
                if kind in ks:
                    self._c[(id(p), kind)] = (tag, t, weakref.ref(p))

    def w(self, p, prec):
        if prec == "fp32":
            return p.detach()
        return self._get(p, "w16", lambda t: t.to(torch.bfloat16).contiguous())

    def wt(self, p, prec):
        """Transposed operand copy [in_features, out_features] in the compute dtype: turns dX = dY . W into the row-major
        (direct-to-LDS) GEMM form; refreshed when the parameter changes (once per optimizer step)."""
        dt = torch.bfloat16 if prec == "bf16" else torch.float32
        def make(t):
            out = torch.empty((t.shape[1], t.shape[0]), dtype=dt, device=t.device)  # fresh row-major strides even for size-1 dims
            out.copy_(t.t())
            return out
        return self._get(p, "wt" + prec, make)

    def b(self, p, prec):
        if p is None:
            return None
        if prec == "fp32":
            return p.detach()
        return self._get(p, "b16", lambda t: t.to(torch.bfloat16).to(torch.float32).contiguous())


_CU_CACHE = {}


def cu_from_lens(lens, device):
    """int32 cu_seqlens of a packed stream on the device.  Cached per (lengths, device) - a training loop asks for the same few every step -
    and uploaded without a host stall (ops.h2d).  The tensors are read-only for every consumer."""
    key = (tuple(int(l) for l in lens), str(device))
    hit = _CU_CACHE.get(key)
    if hit is not None:
        return hit
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(key[0], dtype=torch.int32).cumsum(0)
    out = ops.h2d(cu, device)
    if len(_CU_CACHE) >= 256:
        _CU_CACHE.clear()
    _CU_CACHE[key] = out
    return out


def linear(x32, xb, lin_w, lin_b, prec, wc, residual=None, gelu=False, out_dtype=None):
    """One nn.Linear on a packed stream.  Picks the bf16 copy of the activation in bf16 mode."""
    bf = prec == "bf16"
    a = xb if bf else x32
    if bf and a is None:
        a = ops.cast_bf16(x32)
    if out_dtype is None:
        out_dtype = torch.float32 if residual is not None else (torch.bfloat16 if bf else torch.float32)
    return ops.gemm_nt(a, wc.w(lin_w, prec), wc.b(lin_b, prec), residual=residual, out_dtype=out_dtype, gelu=gelu, round_bf16=bf)


def encoder_stack(stack, x32, xb, cu, max_len, num_heads, prec, wc):
    """nn.TransformerEncoder (post-LN, GELU) on a packed stream.  x32 (M,E) fp32 residual stream, xb its bf16 copy (bf16 mode)."""
    bf = prec == "bf16"
    E = x32.shape[1]
    dh = E // num_heads
    for layer in stack.layers:
        sa = layer.self_attn
        qkv = linear(x32, xb, sa.in_proj_weight, sa.in_proj_bias, prec, wc)
        attn = ops.attn_varlen(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cu, cu, num_heads, dh, max_len)
        y = linear(None, attn, sa.out_proj.weight, sa.out_proj.bias, prec, wc, residual=x32) if bf else \
            linear(attn, None, sa.out_proj.weight, sa.out_proj.bias, prec, wc, residual=x32)
        x32, xb = ops.layernorm(y, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps, want_bf16=bf)
        h = linear(x32, xb, layer.linear1.weight, layer.linear1.bias, prec, wc, gelu=True)
        y = linear(None, h, layer.linear2.weight, layer.linear2.bias, prec, wc, residual=x32) if bf else \
            linear(h, None, layer.linear2.weight, layer.linear2.bias, prec, wc, residual=x32)
        x32, xb = ops.layernorm(y, layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps, want_bf16=bf)
    if stack.norm is not None:
        x32, xb = ops.layernorm(x32, stack.norm.weight.detach(), stack.norm.bias.detach(), stack.norm.eps, want_bf16=bf)
    return x32, xb


def pad_rows(packed, lens, fill_row=None):
    """packed (M,E) fp32 -> (B, Lmax, E) through one row-gather launch; padded rows take `fill_row` (default zeros)."""
    B, Lm, E = len(lens), max(lens), packed.shape[1]
    M = packed.shape[0]
    ext = torch.empty(M + 1, E, dtype=torch.float32, device=packed.device)
    ext[:M] = packed
    if fill_row is None:
        ext[M].zero_()
    else:
        ext[M] = fill_row
    idx = torch.full((B, Lm), M, dtype=torch.int32)
    mask = torch.ones(B, Lm, dtype=torch.bool)
    o = 0
    for b, l in enumerate(lens):
        idx[b, :l] = torch.arange(o, o + l, dtype=torch.int32)
        mask[b, :l] = False
        o += l
    out = ops.gather_rows(ext, idx.reshape(-1).to(packed.device))
    return out.view(B, Lm, E), mask.to(packed.device)


def unpad_rows(padded, mask):
    """(B, Lmax, E) + mask (True = padding, suffix-shaped as create_attention_mask makes it) -> packed (M,E), lens."""
    B, Lm, E = padded.shape
    if mask is None:
        return padded.reshape(B * Lm, E).float().contiguous(), [Lm] * B
    lens = (~mask).sum(dim=1).tolist()
    m = mask.cpu()
    for b, l in enumerate(lens):
        if bool(m[b, :l].any()):
            raise ValueError("latent_attention_mask must mark a suffix of each row as padding (as Encoder.create_attention_mask does)")
    idx = torch.cat([torch.arange(b * Lm, b * Lm + l, dtype=torch.int32) for b, l in enumerate(lens)]).to(padded.device)
    return ops.gather_rows(padded.reshape(B * Lm, E).float().contiguous(), idx), lens


class DecodeEngine:
    """Owns the KV caches, workspaces and hipGraphs of one CachedTransformerDecoder-equivalent."""

    # keys per cross-attention workgroup (split over the memory, merged in the launch).  1024 (4 splits of S = 4096, 512 workgroups) measured
    # +2.8 % tokens/s over 512 on the same box; 256 and 2048 are slower.  ACAI_CROSS_CHUNK overrides (A/B aid).
    CROSS_CHUNK = int(os.environ.get("ACAI_CROSS_CHUNK", "1024"))
    CROSS_SLOTS = 512   # cross-attention workgroups resident at once (two per CU on 256 CUs)

    @classmethod
    def pick_cross_chunk(cls, lens, H):
        """Keys per cross-attention workgroup for this batch.  A uniform batch of 8 x 4096 gives 4 x 8 x 16 = 512 workgroups of 1024 keys: one
        full round of the chip.  A RAGGED batch does not: config 4 (1024 ... 9216 patches) makes 624 workgroups of 1024 keys - a second, mostly
        empty round.  The chunk is therefore the multiple of 64 that minimises rounds x keys per workgroup (ties go to CROSS_CHUNK): 1344 keys
        = 512 workgroups for config 4.  ACAI_CROSS_CHUNK pins it (A/B aid)."""
        if "ACAI_CROSS_CHUNK" in os.environ or not lens:
            return cls.CROSS_CHUNK
        best, best_cost = cls.CROSS_CHUNK, None
        for c in [cls.CROSS_CHUNK] + list(range(512, 4096 + 1, 64)):
            n = sum(-(-l // c) for l in lens) * H
            cost = -(-n // cls.CROSS_SLOTS) * (min(c, max(lens)) + 72)   # + ~72 keys' worth of per-workgroup cost (512- against 1024-key chunks on 8 x 4096: -2.8 % tokens/s)
            if best_cost is None or cost < best_cost:
                best, best_cost = c, cost
        return best

    def __init__(self, blocks, omr, max_batch_size, max_len, prec, device):
        self.blocks = blocks        # CachedTransformerDecoder mirror (layers, norm): parameters are read from it
        self.omr = omr              # OMRDecoder mirror (embedding, positions, unembed) or None
        self.prec = prec
        self.bf = prec == "bf16"
        self.cdt = torch.bfloat16 if self.bf else torch.float32
        self.device = device
        self.Bmax, self.Tmax = int(max_batch_size), int(max_len)
        self.L = len(blocks.layers)
        self.E = blocks.layers[0].hidden_dim
        self.H = blocks.layers[0].num_heads
        self.dh = self.E // self.H
        es = 2 if self.bf else 4
        dhp = max(16 // es, 1)
        while dhp < self.dh:
            dhp *= 2
        self.dhp = dhp
        self.F = blocks.layers[0].linear1.out_features
        self.V = omr.vocab_size if omr is not None else 1
        self.wc = WeightCache()
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)  # noqa: E731
        self.k_self = [z(self.Bmax, self.H, self.Tmax, dhp, dt=self.cdt) for _ in range(self.L)]
        self.v_self = [z(self.Bmax, self.H, self.Tmax, dhp, dt=self.cdt) for _ in range(self.L)]
        self.step = z(2, dt=torch.int32)
        self.finished = z(self.Bmax + 1, dt=torch.int32)
        self.seqs = z(self.Bmax, self.Tmax, dt=torch.int64)
        self.logprobs = z(self.Bmax, self.Tmax)
        self.cross_off = z(self.Bmax, dt=torch.int64)
        self.cross_len = z(self.Bmax, dt=torch.int32)
        self.ws = dict(x=z(self.Bmax, self.E), xn=z(self.Bmax, self.E), qkv=z(self.Bmax, 3 * self.E), attn=z(self.Bmax, self.E),
                       proj=z(self.Bmax, self.E), hid=z(self.Bmax, self.F), logits=z(self.Bmax, self.V))
        self.stats = z(6 * self.Bmax)
        self.tickets = z(self.Bmax * self.H, dt=torch.int32)   # self-resetting arrival counters (in-launch split merge)
        # self-attention: one workgroup per (sequence, head) walks the whole cache (t <= 1536 keys) and writes the output
        # itself - no split, no combine launch
        self.SELF_CHUNK = self.Tmax
        self.self_nsplit = 1
        self.partial = None
        self.k_cross = self.v_cross = None
        self.cross_cap = 0
        self.graphs = {}
        self.stream = torch.cuda.Stream(device=device)  # capture / replay stream (the legacy default stream cannot capture)
        self.B = 0
        self.lens = None
        self.group = 1          # decode rows per stored cross K/V (GRPO rollout groups)
        self._sampler = None    # (top_k, temperature) while a sampling rollout runs, else greedy
        self.uniforms = None    # (Bmax, Tmax) uniforms of the sampling step, allocated on first use
        self.cache_len = 0
        self._desc = None
        self._keep = None

    # ---- cross K/V prefill (MemoryCache.cache_memory_keys_and_vals, kv_caching.py:235-253) -----------------------------
    def prepare(self, mem32, memb, lens, group_size=1):
        """mem32 / memb: packed memory (M, E) fp32 / bf16 copy; lens: per-memory lengths.  group_size > 1: every memory serves `group_size`
        consecutive decode rows (the rollouts of one image, models.py:883-891) - its cross K/V is projected and stored ONCE and the rows'
        offsets alias it, instead of the reference's group_size materialised copies."""
        G = int(group_size)
        B = len(lens) * G
        if B > self.Bmax:
            raise ValueError(f"The current cache has been setup with a max batch size of {self.Bmax}, but found new key tensors with batch size {B}!")
        E, H, dhp, dev = self.E, self.H, self.dhp, self.device
        total = sum(lens) * H * dhp
        if total > self.cross_cap:
            self.cross_cap = total
            self.k_cross = [torch.zeros(total, dtype=self.cdt, device=dev) for _ in range(self.L)]
            self.v_cross = [torch.zeros(total, dtype=self.cdt, device=dev) for _ in range(self.L)]
            self.graphs.clear()  # pointers changed
        offs, o = [], 0
        for l in lens:
            offs.append(o)
            o += l * H * dhp
        self.cross_off[:B] = torch.tensor(offs, dtype=torch.int64).repeat_interleave(G)
        self.cross_len[:B] = torch.tensor(lens, dtype=torch.int32).repeat_interleave(G)
        # the prefill scatters by MEMORY index: its own (ungrouped) offset / length tables
        pre_off, pre_len = torch.tensor(offs, dtype=torch.int64).to(dev), torch.tensor(lens, dtype=torch.int32).to(dev)
        row_seq = torch.cat([torch.full((l,), b, dtype=torch.int32) for b, l in enumerate(lens)]).to(dev)
        row_pos = torch.cat([torch.arange(l, dtype=torch.int32) for l in lens]).to(dev)
        mem = memb if self.bf else mem32
        if mem is None:
            mem = ops.cast_bf16(mem32)
        for i, layer in enumerate(self.blocks.layers):
            ca = layer.multihead_attn
            w = self.wc.w(ca.in_proj_weight, self.prec)[E:]
            b = self.wc.b(ca.in_proj_bias, self.prec)[E:]
            ops.cross_kv_prefill(mem, w, b, row_seq, row_pos, pre_off, pre_len, self.k_cross[i], self.v_cross[i],
                                 H, self.dh, dhp, round_bf16=self.bf)
        self.B, self.lens, self.group = B, [l for l in lens for _ in range(G)], G
        self.cross_chunk = self.pick_cross_chunk(lens, H) if G == 1 else self.CROSS_CHUNK
        self.cross_nsplit = max(1, -(-max(lens) // self.cross_chunk))
        need = B * H * max(self.cross_nsplit, self.self_nsplit) * (dhp + 2)
        if self.partial is None or self.partial.numel() < need:
            self.partial = torch.empty(self.Bmax * H * max(self.cross_nsplit, self.self_nsplit) * (dhp + 2), dtype=torch.float32, device=dev)
            self.graphs.clear()
        self.reset_self_cache()
        self._build_desc()

    def reset_self_cache(self):
        """KVCache.reset (kv_caching.py:47-51): position back to 0 (stale entries are never read: length is step[1]+1)."""
        self.step.zero_()
        self.cache_len = 0

    def _build_desc(self):
        P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        wc, prec, E = self.wc, self.prec, self.E
        layers = (_lib.AcaiDecLayer * self.L)()
        keep = []
        for i, ly in enumerate(self.blocks.layers):
            sa, ca = ly.self_attn, ly.multihead_attn
            t = dict(self_in_w=wc.w(sa.in_proj_weight, prec), self_in_b=wc.b(sa.in_proj_bias, prec),
                     self_out_w=wc.w(sa.out_proj.weight, prec), self_out_b=wc.b(sa.out_proj.bias, prec),
                     cross_q_w=wc.w(ca.in_proj_weight, prec)[:E], cross_q_b=wc.b(ca.in_proj_bias, prec)[:E],
                     cross_out_w=wc.w(ca.out_proj.weight, prec), cross_out_b=wc.b(ca.out_proj.bias, prec),
                     lin1_w=wc.w(ly.linear1.weight, prec), lin1_b=wc.b(ly.linear1.bias, prec),
                     lin2_w=wc.w(ly.linear2.weight, prec), lin2_b=wc.b(ly.linear2.bias, prec),
                     n1_w=ly.norm1.weight.detach(), n1_b=ly.norm1.bias.detach(), n2_w=ly.norm2.weight.detach(), n2_b=ly.norm2.bias.detach(),
                     n3_w=ly.norm3.weight.detach(), n3_b=ly.norm3.bias.detach(),
                     k_self=self.k_self[i], v_self=self.v_self[i], k_cross=self.k_cross[i], v_cross=self.v_cross[i])
            for k, v in t.items():
                assert v.is_contiguous() or k.startswith("cross_q"), k
                setattr(layers[i], k, P(v))
            keep.append(t)
        own, nrm = self.omr, self.blocks.norm
        d = _lib.AcaiDecoder()
        d.B, d.E, d.H, d.dh, d.dhp, d.F, d.V, d.L, d.Tmax = self.B, E, self.H, self.dh, self.dhp, self.F, self.V, self.L, self.Tmax
        d.dtype = _lib.ACAI_BF16 if self.bf else _lib.ACAI_F32
        d.flags = _lib.GEMM_ROUND_BF16 if self.bf else 0
        d.max_len = self.Tmax
        d.cross_group = self.group
        d.self_chunk, d.cross_chunk, d.self_nsplit, d.cross_nsplit = self.SELF_CHUNK, getattr(self, "cross_chunk", self.CROSS_CHUNK), self.self_nsplit, self.cross_nsplit
        d.layers = ctypes.cast(layers, ctypes.POINTER(_lib.AcaiDecLayer))
        top = {}
        if nrm is not None:
            top.update(fn_w=nrm.weight.detach(), fn_b=nrm.bias.detach())
        if own is not None:
            d.bos, d.pad, d.eos = own.bos_idx, own.pad_idx, own.eos_idx
            top.update(emb=own.vocab_embedding.weight.detach(), pos=own.pos_embedding.detach(),
                       unembed_w=wc.w(own.unembed.weight, prec), unembed_b=wc.b(own.unembed.bias, prec))
        for k, v in top.items():
            assert v.is_contiguous(), k
            setattr(d, k, P(v))
        d.cross_off, d.cross_len = P(self.cross_off), P(self.cross_len)
        d.seqs, d.logprobs, d.step, d.finished = P(self.seqs), P(self.logprobs), P(self.step), P(self.finished)
        for k, v in self.ws.items():
            setattr(d, k, P(v))
        d.partial = P(self.partial)
        d.stats = P(self.stats)
        d.tickets = P(self.tickets)
        sig = tuple(getattr(layers[i], f) for i in range(self.L) for f, _ in _lib.AcaiDecLayer._fields_) + \
            tuple(getattr(d, f) for f, t in _lib.AcaiDecoder._fields_ if t is ctypes.c_void_p)
        if getattr(self, "_sig", None) != sig:
            self.graphs.clear()  # a captured graph holds the old pointers
            self._sig = sig
        self._desc, self._keep = d, (layers, keep, top)

    # ---- CachedTransformerDecoder.cached_generate (kv_caching.py:292-302): hidden state for a caller-supplied embedding --
    def hidden_step(self, x):
        if self.cache_len + 1 > self.Tmax:
            raise AssertionError("KV cache overflow: cache_pos + seq_len exceeds max_seq_len")
        x = x.reshape(-1, self.E).to(device=self.device, dtype=torch.float32).contiguous()
        assert x.shape[0] == self.B, f"batch changed from {self.B} to {x.shape[0]} without prepare_caches()"
        _lib.check(_lib.lib().acai_decode_hidden(ctypes.byref(self._desc), x.data_ptr(), ops._st()), "acai_decode_hidden")
        self._x_valid = False   # d->x now holds the caller's embedding, not the chained step's next input
        self.cache_len += 1
        return self.ws["xn"][:self.B]

    # ---- OMRDecoder.cached_generate (models.py:518-528): logits for caller-supplied tokens -----------------------------
    def logits_step(self, tokens, time_step):
        if self.cache_len + 1 > self.Tmax:
            raise AssertionError("KV cache overflow: cache_pos + seq_len exceeds max_seq_len")
        tok = tokens.reshape(-1).to(device=self.device, dtype=torch.int64).contiguous()
        assert tok.numel() == self.B, f"batch changed from {self.B} to {tok.numel()} without prepare_caches()"
        _lib.check(_lib.lib().acai_decode_logits(ctypes.byref(self._desc), tok.data_ptr(), int(time_step), ops._st()), "acai_decode_logits")
        self._x_valid = False
        self.cache_len += 1
        return self.ws["logits"][:self.B]

    # ---- ViTOMR.cached_greedy_generate (models.py:600-615) ----------------------------------------------------------
    def greedy(self, max_len, poll=16, use_graph=True, on_chunk=None):
        """Runs up to max_len-1 greedy steps; returns views seqs (B,max_len) int64 and logprobs (B,max_len) fp32.
        Early exit when every row has produced <eos> (checked every `poll` steps; overshoot is masked later)."""
        B, own = self.B, self.omr
        if max_len > self.Tmax:
            raise RuntimeError(f"{max_len} decoding steps is too long for max sequence length of {self.Tmax}")
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self._greedy_on_stream(max_len, poll, use_graph, on_chunk)
        cur.wait_stream(self.stream)
        return out

    # ---- GRPOViTOMR.cached_forward_rollout_policy (models.py:988-1049) ---------------------------------------------------------
    def sample(self, max_actions, top_k, temperature, uniforms=None, poll=16, use_graph=True):
        """Up to max_actions-1 sampling steps (top-k, temperature, inverse-CDF draw from `uniforms` (B, max_actions) in [0,1) - drawn from
        torch's generator when None).  Returns views seqs (B, max_actions), logprobs (B, max_actions) and the number of steps run."""
        if max_actions > self.Tmax:
            raise RuntimeError(f"{max_actions} decoding steps is too long for max sequence length of {self.Tmax}")
        B = self.B
        if self.uniforms is None:
            self.uniforms = torch.zeros(self.Bmax, self.Tmax, dtype=torch.float32, device=self.device)
        if uniforms is None:
            uniforms = torch.rand(B, max_actions, device=self.device)
        assert uniforms.shape == (B, max_actions)
        self.uniforms[:B, :max_actions] = uniforms.to(device=self.device, dtype=torch.float32)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        self._sampler = (int(top_k), float(temperature))
        try:
            with torch.cuda.stream(self.stream):
                out = self._greedy_on_stream(max_actions, poll, use_graph, None)
        finally:
            self._sampler = None
        cur.wait_stream(self.stream)
        return out

    def greedy_chunks(self, max_len, chunk):
        """Generator over the greedy loop in chunks of `chunk` tokens (streamed inference): yields (tokens_done, all_finished)
        after each chunk; the decode graph is replayed on the engine's stream, the caller's stream waits for it."""
        if max_len > self.Tmax:
            raise RuntimeError(f"{max_len} decoding steps is too long for max sequence length of {self.Tmax}")
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.arm(self.B)
            self.ensure_graph(1)
            self.arm(self.B)
        done, total = 0, max_len - 1
        while done < total:
            n = min(chunk, total - done)
            with torch.cuda.stream(self.stream):
                self.launch_steps(n)
                fin = int(self.finished[self.B].item()) == 0
            cur.wait_stream(self.stream)
            done += n
            self.cache_len = done
            yield done, fin
            if fin:
                return

    def arm(self, B):
        own = self.omr
        self.seqs[:B].fill_(own.pad_idx)
        self.seqs[:B, 0] = own.bos_idx
        self.logprobs[:B].zero_()
        self.finished.zero_()
        self.reset_self_cache()
        self.step.copy_(torch.tensor([1, 0], dtype=torch.int32))
        # input of the first step (<bos> at position 1, quirk Q1); each step's argmax / sampling kernel writes the next step's input
        _lib.check(_lib.lib().acai_decode_embed(ctypes.byref(self._desc), ops._st()), "acai_decode_embed")
        self._x_valid = True

    STEPS_PER_GRAPH = 8   # a graph replay costs ~10-15 us of launch latency: amortise it over several decode steps

    def _step(self, st):
        """One decode step on the current stream: greedy, or (self._sampler = (top_k, temperature)) a sampling step."""
        smp = self._sampler
        if smp is None:
            _lib.check(_lib.lib().acai_decode_step(ctypes.byref(self._desc), st), "acai_decode_step")
        else:
            _lib.check(_lib.lib().acai_decode_sample_step(ctypes.byref(self._desc), self.uniforms.data_ptr(), int(smp[0]), float(smp[1]), st),
                       "acai_decode_sample_step")

    def ensure_graph(self, nsteps=1):
        """hipGraph of `nsteps` consecutive decode steps for the current (B, cross split) configuration.  Must run on self.stream."""
        B = self.B
        key = (B, self.cross_nsplit, getattr(self, "cross_chunk", self.CROSS_CHUNK), nsteps, self._sampler, self.group)
        g = self.graphs.get(key)
        if g is None:
            st = ops._st()
            # warm-up launch outside capture (first-use code-object load must not happen inside a capture), then re-arm
            self._step(st)
            torch.cuda.current_stream().synchronize()
            self.arm(B)
            torch.cuda.current_stream().synchronize()
            g = ops.Graph()
            g.begin()
            try:
                for _ in range(nsteps):
                    self._step(st)
            finally:
                g.end()
            self.graphs[key] = g
        return g

    def launch_steps(self, n, use_graph=True):
        """Enqueue n decode steps on the current stream (graphs of STEPS_PER_GRAPH steps + single-step graphs for the rest).
        A chained step takes its input embedding from d->x, which the PREVIOUS step's argmax / sampling kernel wrote (acai_decode_step no
        longer embeds by itself).  The stepwise entry points (logits_step / hidden_step) overwrite d->x; after one of them the input is
        rebuilt from the device-side sequence state (`acai_decode_embed`: seqs[:, t-1] at position t) before the chain goes on."""
        if n > 0 and not getattr(self, "_x_valid", False):
            _lib.check(_lib.lib().acai_decode_embed(ctypes.byref(self._desc), ops._st()), "acai_decode_embed")
            self._x_valid = True
        if not use_graph:
            st = ops._st()
            for _ in range(n):
                self._step(st)
            return
        big = self.STEPS_PER_GRAPH
        while n >= big:
            self.ensure_graph(big).launch()
            n -= big
        while n > 0:
            self.ensure_graph(1).launch()
            n -= 1

    def _greedy_on_stream(self, max_len, poll, use_graph, on_chunk):
        B = self.B
        self.arm(B)
        if use_graph:   # capture (and warm up) before the loop, then re-arm: the warm-up launch advances the device state
            self.ensure_graph(1)
            self.ensure_graph(self.STEPS_PER_GRAPH)
            self.arm(B)
        done = 0
        total = max_len - 1
        while done < total:
            n = min(poll, total - done)
            self.launch_steps(n, use_graph)
            done += n
            self.cache_len = done
            if on_chunk is not None:
                on_chunk(done)
            if int(self.finished[B].item()) == 0:  # device -> host sync once per `poll` tokens
                break
        return self.seqs[:B, :max_len], self.logprobs[:B, :max_len], done
