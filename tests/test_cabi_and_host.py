"""CPU-side checks (no GPU): the C-ABI library builds/loads and exports every symbol include/acai_omr_hip.h declares;
host logic of the mirror modules (state_dict compatibility with reference checkpoints, error behaviour, integer
bookkeeping)."""
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT, VOCAB, load_golden


def test_library_exports_every_declared_symbol():
    from acai_omr_amd import _lib
    _lib.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "acai_omr_hip.h")).read()
    declared = set(re.findall(r"\b(acai_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    for name in declared:
        assert getattr(L, name) is not None
    assert L.acai_version() == 1


def test_struct_layouts_match_header():
    """ctypes mirrors of AcaiDecLayer / AcaiDecoder: field order and count as in the header."""
    from acai_omr_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "acai_omr_hip.h")).read()
    body = hdr[hdr.index("typedef struct {"):hdr.index("} AcaiDecLayer;")]
    names = re.findall(r"\*\s*([a-z0-9_]+)\s*[;,]", body)
    names += []
    assert [f for f, _ in _lib.AcaiDecLayer._fields_] == names
    import ctypes
    assert ctypes.sizeof(_lib.AcaiDecLayer) == 8 * len(names)
    assert ctypes.sizeof(_lib.AcaiDecoder) == 20 * 4 + 8 * 23
    assert ctypes.sizeof(_lib.AcaiAdamWTensor) == 56 and ctypes.sizeof(_lib.AcaiAdamWGroup) == 32   # include/acai_omr_hip.h: AcaiAdamWTensor / AcaiAdamWGroup
    assert ctypes.sizeof(_lib.AcaiCastEntry) == 48 and [f for f, _ in _lib.AcaiCastEntry._fields_] == ["src", "dst16", "dst16t", "dst32r", "rows", "cols", "tile0", "pad_"]


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly instead of computing on the host."""
    from acai_omr_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-5)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros(2, 8), torch.zeros(4, 8))


def test_state_dict_keys_match_reference_checkpoints():
    from acai_omr_amd.models.models import MAE, FineTuneOMREncoder, OMRDecoder, TeacherForcedViTOMR
    fx = load_golden("vitomr_small")
    cfg = fx["cfg"]
    enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                             num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"])
    dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"])
    m = TeacherForcedViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"])
    assert set(m.state_dict().keys()) == set(fx["state_dict"].keys())
    m.load_state_dict(fx["state_dict"])
    cached = dec.to_cached_version(4, torch.bfloat16)
    assert set(cached.state_dict().keys()) == set(dec.state_dict().keys())   # caches are non-persistent
    # the reference's own debug checkpoint (debug_pretrained_mae.pth, 61 tensors)
    dbg = load_golden("mae_debug_ckpt")
    mae = MAE(0.75, 16, 60, 200, encoder_hidden_dim=10, decoder_hidden_dim=10, encoder_kwargs=dict(num_layers=2, num_heads=1, mlp_dim=1),
              decoder_kwargs=dict(num_layers=2, num_heads=1, mlp_dim=1))
    assert len(dbg["state_dict"]) == 61
    mae.load_state_dict(dbg["state_dict"])
    # MAE -> OMR encoder transfer (models.py:679-713) incl. frozen / fine-tune split and renumbering
    enc2 = FineTuneOMREncoder(16, 60, 200, 1, num_layers=2, hidden_dim=10, num_heads=1, mlp_dim=1)
    dec2 = OMRDecoder(16, VOCAB, num_layers=1, hidden_dim=8, num_heads=1, mlp_dim=4)
    tf = TeacherForcedViTOMR(enc2, dbg["state_dict"], dec2, transition_head_dim=6)
    assert torch.equal(tf.encoder.frozen_blocks.layers[0].linear1.weight, dbg["state_dict"]["encoder.encoder_blocks.layers.0.linear1.weight"])
    assert torch.equal(tf.encoder.fine_tune_blocks.layers[0].linear1.weight, dbg["state_dict"]["encoder.encoder_blocks.layers.1.linear1.weight"])
    assert torch.equal(tf.encoder.fine_tune_blocks.norm.weight, dbg["state_dict"]["encoder.encoder_blocks.norm.weight"])
    # freezing rules (models.py:667-677)
    assert not tf.encoder.pos_embedding.requires_grad and not tf.encoder.projection.weight.requires_grad
    assert all(not p.requires_grad for p in tf.encoder.frozen_blocks.parameters())
    assert all(p.requires_grad for p in tf.encoder.fine_tune_blocks.parameters())
    groups, lrs = tf.create_fine_tune_param_groups(1e-4, 1e-5, 0.9)
    assert len(groups) == 2 + 1 + 3 and lrs == [1e-5]


def test_host_bookkeeping_kats():
    from acai_omr_amd.models.models import OMRDecoder, ViTOMR, batchify_and_split_lmx_seqs
    inp, tgt, mask = batchify_and_split_lmx_seqs((torch.tensor([0, 2, 3, 226]), torch.tensor([0, 2, 2, 3, 4, 226])), 1, "cpu")
    assert inp.tolist() == [[0, 2, 3, 226, 1], [0, 2, 2, 3, 4]] and tgt.tolist() == [[2, 3, 226, 1, 1], [2, 2, 3, 4, 226]]
    assert mask.int().tolist() == [[0, 0, 0, 0, 1], [0, 0, 0, 0, 0]]
    dec = OMRDecoder(8, VOCAB, num_layers=1, hidden_dim=8, num_heads=1, mlp_dim=4)
    assert (dec.bos_idx, dec.pad_idx, dec.eos_idx, dec.vocab_size) == (0, 1, 2, 227)
    # PrepareLMXSequence KATs (tests/test_omr_teacher_force_train.py:22-28) are vocabulary lookups
    assert [dec.tokens_to_idxs[t] for t in "measure key:fifths:-7 time".split()] == [3, 4, 19]
    assert [dec.tokens_to_idxs[t] for t in "tremolo:4 C1".split()] == [226, 66]
    v = ViTOMR(None, None, dec)
    assert v.create_inference_mask(torch.tensor([[0, 2, 10, 2], [0, 20, 20, 2]])).int().tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]
    s, lp, m = v.mask_and_clip_seqs(torch.tensor([[0, 5, 2, 7, 7], [0, 2, 9, 9, 9]]), torch.ones(2, 5))
    assert s.tolist() == [[0, 5, 2], [0, 2, 1]] and lp.tolist() == [[1, 1, 1], [1, 1, 0]] and m.tolist() == [[True, True, True], [True, True, False]]
    with pytest.raises(RuntimeError):
        dec.prepare_caches(torch.zeros(1, 2, 8))
    with pytest.raises(ValueError):
        dec.forward(torch.zeros(1, 9, dtype=torch.long), torch.zeros(1, 2, 8), None, None)


def test_host_helpers_stringify_and_schedules():
    """stringify_lmx_seq (utils.py:194-202) and the warm-up + cosine schedules (utils.py:204-222) drive FusedAdamW's param groups exactly as
    they drive torch.optim.AdamW's (no step() is taken here: that needs the GPU)."""
    import warnings
    import torch
    from acai_omr_amd.optim import FusedAdamW
    from acai_omr_amd.utils import cosine_anneal_with_warmup, ragged_collate_fn, stepwise_cosine_anneal_with_warmup, stringify_lmx_seq
    vocab = {0: "<bos>", 1: "<pad>", 2: "<eos>", 3: "measure", 4: "key:fifths:-7", 19: "time"}
    assert stringify_lmx_seq(torch.tensor([0, 3, 4, 19, 2]), vocab) == "measure key:fifths:-7 time"
    assert stringify_lmx_seq(torch.tensor([0, 3, 4]), vocab) == "measure key:fifths:-7"      # truncated: no <eos>
    assert ragged_collate_fn([(1, 2), (3, 4)]) == [(1, 2), (3, 4)]

    def lrs(opt_cls, make):
        ps = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))]
        opt = opt_cls([dict(params=ps[:1], lr=1e-3), dict(params=ps[1:], lr=1.5e-4)], betas=(0.9, 0.95), weight_decay=0.05)
        sch = make(opt)
        out = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(12):
                out.append([g["lr"] for g in opt.param_groups])
                sch.step()
        return out

    for make in (lambda o: cosine_anneal_with_warmup(o, 2, 10, 1e-6), lambda o: cosine_anneal_with_warmup(o, 1, 3, 1e-6, num_train_batches=4),
                 lambda o: stepwise_cosine_anneal_with_warmup(o, 3, 2, 1e-6, 6)):
        a, b = lrs(torch.optim.AdamW, make), lrs(FusedAdamW, make)
        assert a == b and a[0][0] == 1e-3 * 5e-3 and max(x[0] for x in a) <= 1e-3 + 1e-12


def test_grpo_host_helpers_reference_vectors():
    """Known answers of the reference's tests/test_vitomr.py for the GRPO helpers around the rollout decode (rollout masks :414-442, rollout
    preparation :457-497, latent expansion :395-412, state-dict conversion + freezing :376-393) - host logic, no kernels."""
    import torch
    from conftest import VOCAB
    from acai_omr_amd.models.models import FineTuneOMREncoder, GRPOViTOMR, OMRDecoder, TeacherForcedViTOMR
    kw = dict(num_layers=2, num_heads=1, hidden_dim=10, mlp_dim=1)
    torch.manual_seed(0)
    tf = TeacherForcedViTOMR(FineTuneOMREncoder(16, 60, 200, 1, **kw), None, OMRDecoder(1536, VOCAB, **kw))
    g = GRPOViTOMR(tf.encoder, tf.transition_head, tf.decoder, tf.state_dict())
    assert all(not p.requires_grad for p in g.encoder.parameters()) and all(not p.requires_grad for p in g.transition_head.parameters())
    assert all(p.requires_grad for p in g.decoder.parameters())
    assert all(m.p == 0.0 for m in g.encoder.modules() if isinstance(m, torch.nn.Dropout))
    # one encoder stack again: frozen layer 0 + fine-tuned layer 0 -> encoder_blocks.layers.{0,1}
    assert torch.equal(g.encoder.encoder_blocks.layers[0].linear1.weight, tf.encoder.frozen_blocks.layers[0].linear1.weight)
    assert torch.equal(g.encoder.encoder_blocks.layers[1].linear1.weight, tf.encoder.fine_tune_blocks.layers[0].linear1.weight)
    bos, eos, pad = g.decoder.bos_idx, g.decoder.eos_idx, g.decoder.pad_idx
    T, F = True, False
    cases = [([[bos, 10, 10, eos], [bos, 20, 20, eos]], [[T] * 4, [T] * 4]),
             ([[bos, eos, 10, eos], [bos, 20, 20, eos]], [[T, T, F, F], [T] * 4]),
             ([[bos, 10, 10, 10], [bos, 20, 20, 20]], [[T] * 4, [T] * 4]),
             ([[bos, 20, 20, 20], [bos, eos, 10, 10]], [[T] * 4, [T, T, F, F]])]
    for rollouts, expected in cases:
        assert torch.equal(g.create_rollout_mask(torch.tensor(rollouts)), torch.tensor(expected))
    prep = [([[bos, 10, 10, eos], [bos, 10, 10, eos]], [[F] * 3, [F] * 3]),
            ([[bos, 10, 10, eos], [bos, eos, pad, pad]], [[F] * 3, [F, T, T]]),
            ([[bos, 10, eos, pad], [bos, 10, 10, 10]], [[F, F, T], [F] * 3])]
    for rollouts, expected in prep:
        r = torch.tensor(rollouts)
        out, mask = g.prepare_rollouts_for_policy_theta(r, g.create_rollout_mask(r))
        assert torch.equal(out, r[:, :-1]) and torch.equal(mask, torch.tensor(expected))
    lat = torch.arange(2 * 3 * 10, dtype=torch.float32).view(2, 3, 10)
    msk = torch.tensor([[F, F, T], [F, F, F]])
    lx, mx = g.expand_img_latent_for_rollout(lat, msk, 2)
    assert torch.equal(lx, torch.cat([lat[:1].repeat(2, 1, 1), lat[1:].repeat(2, 1, 1)])) and torch.equal(mx, torch.cat([msk[:1].repeat(2, 1), msk[1:].repeat(2, 1)]))


def test_cu_seqlens_are_cached_per_lengths_and_uploads_pass_cpu_through():
    """engine.cu_from_lens: int32 prefix sums, one tensor per (lengths, device) - a training loop asks for the same few every step - and
    ops.h2d leaves a CPU destination alone (on a GPU it stages through pinned memory and copies without stalling the host)."""
    from acai_omr_amd import engine, ops
    a = engine.cu_from_lens([3, 5, 1], "cpu")
    assert a.dtype == torch.int32 and a.tolist() == [0, 3, 8, 9]
    assert engine.cu_from_lens((3, 5, 1), "cpu") is a and engine.cu_from_lens([3, 5, 2], "cpu") is not a
    t = torch.arange(6, dtype=torch.int64)
    assert ops.h2d(t, "cpu") is t and ops.h2d(t, "cpu", torch.int32).dtype == torch.int32


def test_attn_backward_workspace_query_matches_the_dispatch():
    """acai_attn_varlen_bwd_workspace_bytes names the calls the one-pass form takes: [total_q + 64][H][32] fp32 for them, zero for the rest.  A
    host-side function: no device is touched."""
    from acai_omr_amd import _lib
    L = _lib.lib()
    bf, f32 = _lib.ACAI_BF16, _lib.ACAI_F32
    q = lambda *a: int(L.acai_attn_varlen_bwd_workspace_bytes(*a))
    full = (16384 + 64) * 16 * 32 * 4
    #            B  H  dh  max_q max_k total_q total_k flags dtype p  pre
    assert q(4, 16, 32, 4096, 4096, 16384, 16384, 0, bf, 0.0, 1) == full
    assert q(4, 16, 32, 4096, 4096, 16000, 16384, 0, bf, 0.0, 1) == (16000 + 64) * 16 * 32 * 4   # ragged queries
    assert q(4, 16, 32, 4096, 4000, 16384, 15000, 0, bf, 0.0, 1) == full   # ragged keys, not a multiple of 512: the partial-block launch
    assert q(4, 16, 32, 4096, 4096, 16384, 16384, 0, bf, 0.0, 0) == 0      # q not prescaled
    assert q(4, 16, 32, 4096, 4096, 16384, 16384, 1, bf, 0.0, 1) == 0      # causal
    assert q(4, 16, 32, 4096, 4096, 16384, 16384, 2, bf, 0.0, 1) == 0      # accumulating dk / dv
    assert q(4, 16, 32, 4096, 4096, 16384, 16384, 0, bf, 0.1, 1) == 0      # dropout
    assert q(4, 16, 32, 4096, 4096, 16384, 16384, 0, f32, 0.0, 1) == 0     # fp32
    assert q(4, 16, 64, 4096, 4096, 16384, 16384, 0, bf, 0.0, 1) == 0      # d_h = 64
    assert q(4, 16, 32, 256, 512, 1024, 2048, 0, bf, 0.0, 1) == 0          # short query side: the two-kernel form
    assert q(4, 16, 32, 512, 400, 2048, 1600, 0, bf, 0.0, 1) == 0          # short key side


def test_asm_checks_flag_what_they_are_for():
    """acai_omr_amd/_asmcheck.py runs inside _lib.build() and fails the build: (1) a register written by an inline-asm load the compiler cannot
    see, touched before the kernel's counted vmcnt wait (the miscompile gemm_nt_pp_kernel once hit); (2) more LDS operations behind the staged
    tiles' stores than attn_fwd64w_kernel's counted tile barrier allows.  Doctored snippets: the clean form passes, the broken form is named."""
    from acai_omr_amd import _asmcheck
    head = "_ZN12_GLOBAL__N_117gemm_nt_pp_kernelItLi1EEEvNS_8GemmArgsE:\n"
    clean = head + "\t;;#ASMSTART\n\tglobal_load_dword v7, v3, s[2:3]\n\t;;#ASMEND\n\tv_add_f32 v1, v2, v3\n\ts_waitcnt vmcnt(4)\n\tv_mov_b32 v9, v7\n\ts_endpgm\n.Lfunc_end0:\n"
    broken = head + "\t;;#ASMSTART\n\tglobal_load_dword v7, v3, s[2:3]\n\t;;#ASMEND\n\tv_mov_b32 v9, v7\n\ts_waitcnt vmcnt(4)\n\ts_endpgm\n.Lfunc_end0:\n"
    tracked = head + "\tglobal_load_dword v7, v3, s[2:3]\n\tv_mov_b32 v9, v7\n\ts_endpgm\n.Lfunc_end0:\n"     # a compiler-issued load: its own wait counts cover it
    assert _asmcheck.check_untracked_loads(clean) == ([], 1)
    p, n = _asmcheck.check_untracked_loads(broken)
    assert n == 1 and len(p) == 1 and "v_mov_b32 v9, v7" in p[0]
    assert _asmcheck.check_untracked_loads(tracked) == ([], 0)
    whead = "_ZN12_GLOBAL__N_118attn_fwd64w_kernelILi0EEEv8AttnArgs:\n"
    reads = "".join(f"\tds_read_b64_tr_b16 v[{2 * i}:{2 * i + 1}], v100\n" for i in range(4))
    ok = whead + "\tds_write_b128 v1, v[2:5]\n" + reads + "\ts_waitcnt lgkmcnt(4)\n\ts_barrier\n\ts_endpgm\n.Lfunc_end0:\n"
    bad = whead + "\tds_write_b128 v1, v[2:5]\n" + reads + "\tds_read_b128 v[20:23], v101\n\ts_waitcnt lgkmcnt(4)\n\ts_barrier\n\ts_endpgm\n.Lfunc_end0:\n"
    assert _asmcheck.check_fwd64w_barrier(ok) == ([], 1)
    p, n = _asmcheck.check_fwd64w_barrier(bad)
    assert n == 1 and len(p) == 1 and "5 LDS operations" in p[0]
    assert _asmcheck.check_fwd64w_barrier("nothing here\n")[0]      # the kernel vanished: that is a finding too
    # (3) an inline-asm MFMA whose source register a VALU instruction wrote less than two wait states earlier (round 4: the compiler's tuple copies
    # in front of attn_bwd64w's asm MFMAs gave wrong gradients); an s_nop inside the asm statement, or two other instructions between, clear it;
    # an accumulate chain of MFMAs on the same C / D registers is legal back to back
    bhead = "_ZN12_GLOBAL__N_121attn_bwd64w_dq_kernelE7BwdArgs:\n"
    mf = "\tv_mfma_f32_32x32x16_bf16 a[32:47], a[72:75], v[92:95], a[32:47]\n"
    hazard = bhead + "\tv_accvgpr_mov_b32 a75, a3\n\t;;#ASMSTART\n" + mf + "\t;;#ASMEND\n\ts_endpgm\n.Lfunc_end0:\n"
    padded = bhead + "\tv_accvgpr_mov_b32 a75, a3\n\t;;#ASMSTART\n\ts_nop 1\n" + mf + "\t;;#ASMEND\n\ts_endpgm\n.Lfunc_end0:\n"
    spaced = bhead + "\tv_cvt_pk_bf16_f32 v93, v1, v2\n\tv_exp_f32_e32 v7, v8\n\tds_read_b128 v[20:23], v101\n\t;;#ASMSTART\n" + mf + "\t;;#ASMEND\n\ts_endpgm\n.Lfunc_end0:\n"
    early_exit = bhead + "\ts_endpgm\n.LBB0_2:\n\tv_accvgpr_mov_b32 a75, a3\n\t;;#ASMSTART\n" + mf + "\t;;#ASMEND\n\ts_endpgm\n.Lfunc_end0:\n"   # (an early return in front)
    chain = bhead + "\t;;#ASMSTART\n" + mf + "\t;;#ASMEND\n\t;;#ASMSTART\n" + mf + "\t;;#ASMEND\n\ts_endpgm\n.Lfunc_end0:\n"
    p, n = _asmcheck.check_asm_mfma_operands(hazard)
    assert n == 1 and len(p) == 1 and "v_accvgpr_mov_b32 a75, a3" in p[0]
    assert _asmcheck.check_asm_mfma_operands(padded) == ([], 1)
    assert _asmcheck.check_asm_mfma_operands(spaced) == ([], 1)
    assert _asmcheck.check_asm_mfma_operands(chain) == ([], 2)
    p, n = _asmcheck.check_asm_mfma_operands(early_exit)
    assert n == 1 and len(p) == 1   # (the function's first s_endpgm is not its end)
    assert _asmcheck.check_asm_mfma_operands("nothing here\n")[0]


def test_shared_tensor_gradient_registry_is_scoped_to_one_backward_pass():
    """autograd_path._tgrad_prev / _tgrad_note (round 4: the second consumer of a shared non-leaf tensor adds its gradient into the tensor the first
    consumer returned): host logic only, on the CPU with a toy Function - same gradient as plain autograd, one registration per backward pass,
    nothing reused by a later pass, nothing outside a backward pass."""
    from acai_omr_amd.train import autograd_path as AP

    adds = []

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t, w):
            ctx.save_for_backward(t, w)
            return (t * w).sum()

        @staticmethod
        def backward(ctx, g):
            t, w = ctx.saved_tensors
            prev = AP._tgrad_prev(t)
            adds.append(prev is not None)
            if prev is not None:
                prev.add_(g * w)          # what the kernel's accumulate epilogue does
                return None, None
            out = g * w
            AP._tgrad_note(t, out)
            return out, None

    assert AP._tgrad_prev(torch.ones(3)) is None          # outside a backward pass: never a hit
    x = torch.arange(6.0, requires_grad=True)
    w1, w2 = torch.full((6,), 2.0), torch.full((6,), 5.0)
    for _ in range(2):                                     # two separate passes: the second must not see the first one's tensor
        x.grad = None
        del adds[:]
        t = x * 3.0                                        # the shared non-leaf tensor
        (Consumer.apply(t, w1) + Consumer.apply(t, w2)).backward()
        assert sorted(adds) == [False, True]
        assert torch.equal(x.grad, torch.full((6,), 3.0 * 7.0))
