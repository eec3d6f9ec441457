"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (needs /root/reference, read-only):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python /root/repo/oracle/gen_golden.py

cwd must be /root/reference because the reference opens "lmx_vocab.txt" cwd-relative
(tests/test_kv_caching.py:8).  Nothing of the reference's source travels: a fixture holds
seeded weights (state_dict tensors), inputs and the reference's outputs.  While generating,
every oracle function is also checked against the reference output (asserts below), so a
fixture is only written when oracle == reference.
"""
import os
import sys

import torch
from torch.amp import autocast

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from acai_omr.models.models import (MAE, FineTuneOMREncoder, GRPOViTOMR, MAELoss, OMRCELoss, OMRDecoder, OMREncoder,  # noqa: E402
                                    Encoder, TeacherForcedViTOMR, batchify_and_split_lmx_seqs)
from oracle import vitomr_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
VOCAB = "lmx_vocab.txt"


def sd_cpu(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def maxdiff(a, b):
    return float((a.float() - b.float()).abs().max())


def build_vitomr(cfg, dropout_zero=False):
    enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"],
                             hidden_dim=cfg["enc_dim"], num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"],
                             transformer_dropout=0.0 if dropout_zero else 0.05)
    dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"],
                     num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"], transformer_dropout=0.0 if dropout_zero else 0.1)
    return TeacherForcedViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"],
                               transition_head_dropout=0.0 if dropout_zero else 0.05)


def run_reference_greedy(vitomr, imgs, cfg, use_autocast, cache_dtype):
    """vitomr_inference.inference plumbing (vitomr_inference.py:73-86), logits recorded per step."""
    vitomr.eval()
    cached = vitomr.decoder.to_cached_version(8, cache_dtype)
    cached.load_state_dict(vitomr.decoder.state_dict())
    plain = vitomr.decoder
    vitomr.decoder = cached.eval()
    rec = []
    orig = cached.cached_generate

    def wrapped(token_t, time_step, latent_attention_mask=None):
        out = orig(token_t, time_step, latent_attention_mask)
        rec.append(out.detach().float().squeeze(1).clone())
        return out

    cached.cached_generate = wrapped
    with torch.no_grad():
        lat, mask = vitomr.encoder(imgs)
        if use_autocast:
            with autocast(device_type="cpu", dtype=torch.bfloat16):
                mem = vitomr.transition_head(lat)
                seqs, lps, smask = vitomr.cached_greedy_generate(mem, mask, max_len=cfg["gen_len"])
        else:
            mem = vitomr.transition_head(lat)
            seqs, lps, smask = vitomr.cached_greedy_generate(mem, mask, max_len=cfg["gen_len"])
    vitomr.decoder = plain
    return dict(latent=lat, latent_mask=mask, memory=mem.float(), seqs=seqs, log_probs=lps, seq_mask=smask,
                step_logits=torch.stack(rec, 1))


def gen_vitomr(name, cfg, imgs, seed):
    torch.manual_seed(seed)
    vitomr = build_vitomr(cfg)
    # default init leaves LN weights at 1 / biases at 0 and small logits; perturb so that parity is not vacuous
    with torch.no_grad():
        for n, p in vitomr.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
        vitomr.decoder.unembed.weight.mul_(4.0)
    sd = sd_cpu(vitomr)
    fx = dict(cfg=cfg, state_dict=sd, imgs=imgs)

    ref32 = run_reference_greedy(vitomr, imgs, cfg, use_autocast=False, cache_dtype=torch.float)
    ref16 = run_reference_greedy(vitomr, imgs, cfg, use_autocast=True, cache_dtype=torch.bfloat16)
    fx["ref_fp32"], fx["ref_bf16"] = ref32, ref16

    # ---- oracle vs reference -------------------------------------------------------
    nb = sd["encoder.fine_tune_blocks.norm.bias"]
    lat, mask = O.encoder_forward_padded(imgs, sd, "encoder.", cfg["P"], cfg["enc_heads"], "omr_ft", "fp32",
                                         final_norm_bias_fill=nb if cfg["enc_heads"] % 2 == 0 else None)
    assert torch.equal(mask, ref32["latent_mask"])
    valid = ~mask
    d = maxdiff(lat[valid], ref32["latent"][valid])
    print(f"[{name}] encoder fp32 max|d| = {d:.3e}")
    assert d < 2e-5
    if cfg["enc_heads"] % 2 == 0:  # fast path: padded rows are the final norm's bias
        assert maxdiff(lat, ref32["latent"]) < 2e-5
    packed, lens = O.unpad(lat, mask)
    for prec, ref in (("fp32", ref32), ("bf16", ref16)):
        mem = O.transition_head(packed, sd, prec)
        dm = maxdiff(O.pad_packed(mem, lens)[0][valid], ref["memory"][valid])
        seqs, lps, smask, logits = O.greedy_generate(mem, lens, sd, cfg["dec_heads"], prec, cfg["gen_len"], return_logits=True)
        T = ref["step_logits"].shape[1]
        dl = maxdiff(logits[:, :T], ref["step_logits"])
        print(f"[{name}] {prec}: memory max|d| = {dm:.3e}  step logits max|d| = {dl:.3e}  tokens equal = {torch.equal(seqs, ref['seqs'])}")
        if prec == "fp32":
            assert dm < 2e-5 and dl < 1e-4
            assert torch.equal(seqs, ref["seqs"]) and torch.equal(smask, ref["seq_mask"])
            assert maxdiff(lps, ref["log_probs"]) < 1e-4
        else:
            # autocast restatement: same rounding points, different accumulation order
            assert dm < 0.05 and dl < 0.25, (dm, dl)
        fx[f"oracle_{prec}_tokens_equal"] = bool(torch.equal(seqs, ref["seqs"]))
    torch.save(fx, os.path.join(OUT, name + ".pt"))


def gen_teacher_forced(name, cfg, imgs, seed):
    torch.manual_seed(seed)
    vitomr = build_vitomr(cfg, dropout_zero=True)
    with torch.no_grad():
        for n, p in vitomr.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = sd_cpu(vitomr)
    g = torch.Generator().manual_seed(seed + 1)
    lmx = []
    for L in cfg["lmx_lens"]:
        body = torch.randint(3, 227, (L,), generator=g)
        lmx.append(torch.cat([torch.tensor([0]), body, torch.tensor([2])]))
    batch = list(zip(imgs, lmx))
    vitomr.train()  # dropout p = 0: deterministic, torch slow path (padded + masks)
    pred, tgt = vitomr(batch)
    loss = OMRCELoss(vitomr.decoder.pad_idx)(pred, tgt)
    loss.backward()
    grad_names = ["decoder.unembed.weight", "decoder.decoder_blocks.layers.0.multihead_attn.in_proj_weight",
                  "transition_head.0.weight", "encoder.fine_tune_blocks.layers.0.linear1.weight", "encoder.projection.weight",
                  "encoder.pos_embedding", "decoder.vocab_embedding.weight"]
    params = dict(vitomr.named_parameters())
    grad_names = [n for n in grad_names if params[n].grad is not None]  # frozen params (M:667-677) have none
    grads = {n: params[n].grad.detach().clone() for n in grad_names}
    fx = dict(cfg=cfg, state_dict=sd, imgs=imgs, lmx=lmx, pred=pred.detach(), target=tgt, loss=loss.detach(), grads=grads)

    # reference KAT for batchify_and_split_lmx_seqs (tests/test_vitomr.py:151-172) restated on the oracle
    i2, t2, m2 = batchify_and_split_lmx_seqs(tuple(lmx), 1, "cpu")
    i1, t1, m1 = O.batchify_and_split_lmx_seqs(lmx, 1)
    assert torch.equal(i1, i2) and torch.equal(t1, t2) and torch.equal(m1, m2)

    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    opred, otgt = O.teacher_forced_forward(batch, sdg, cfg["enc_heads"], cfg["dec_heads"], cfg["P"], "fp32")
    oloss = O.ce_loss(opred, otgt)
    oloss.backward()
    valid = tgt != 1
    dp = maxdiff(opred[valid], pred[valid])
    print(f"[{name}] TF pred max|d| = {dp:.3e} loss d = {abs(float(oloss) - float(loss)):.3e}")
    assert dp < 1e-4 and abs(float(oloss) - float(loss)) < 1e-5
    for n in grad_names:
        dg = maxdiff(sdg[n].grad, grads[n])
        print(f"[{name}]   grad {n}: max|d| = {dg:.3e}  (|g|max {float(grads[n].abs().max()):.3e})")
        assert dg < 1e-4 * max(1.0, float(grads[n].abs().max()))
    torch.save(fx, os.path.join(OUT, name + ".pt"))


def gen_mae(name, cfg, imgs, tgts, seed, sd_override=None):
    torch.manual_seed(seed)
    mae = MAE(cfg["mask_ratio"], cfg["P"], cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"],
              decoder_hidden_dim=cfg["dec_dim"], encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"])
    if sd_override is not None:
        mae.load_state_dict(sd_override)
    else:
        with torch.no_grad():
            for n, p in mae.named_parameters():
                if "norm" in n:
                    p.add_(0.1 * torch.randn_like(p))
    sd = sd_cpu(mae)
    batch = list(zip(imgs, tgts))
    # noise injection: the reference draws torch.rand(N_i) once per image, in order (models.py:110)
    torch.manual_seed(seed + 7)
    noises = [torch.rand((im.shape[-2] // cfg["P"]) * (im.shape[-1] // cfg["P"])) for im in imgs]
    torch.manual_seed(seed + 7)
    mae.train()
    pred, loss_mask, target = mae(batch)
    loss = MAELoss()(pred, loss_mask, target)
    loss.backward()
    grad_names = ["encoder.pos_embedding", "mask_token", "decoder_pos_embedding", "encoder.projection.weight",
                  "decoder.decoder_blocks.layers.0.self_attn.in_proj_weight", "decoder_unembed.weight",
                  "encoder.encoder_blocks.layers.0.linear2.weight"]
    params = dict(mae.named_parameters())
    grads = {n: params[n].grad.detach().clone() for n in grad_names}
    fx = dict(cfg=cfg, state_dict=sd, imgs=imgs, tgts=tgts, noises=noises, pred=pred.detach(), loss_mask=loss_mask,
              target=target, loss=loss.detach(), grads=grads)

    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    eh, dh = cfg["enc_kwargs"]["num_heads"], cfg["dec_kwargs"]["num_heads"]
    opred, omask, otgt, lens = O.mae_forward(batch, noises, sdg, cfg["P"], cfg["mask_ratio"], eh, dh)
    oloss = O.mae_loss(opred, omask, otgt)
    oloss.backward()
    ppred, pmask = O.pad_packed(opred, lens)
    valid = ~pmask
    assert torch.equal(O.pad_packed(omask, lens, False)[0], loss_mask)
    assert torch.equal(O.pad_packed(otgt, lens)[0], target)
    dp = maxdiff(ppred[valid], pred[valid])
    print(f"[{name}] MAE pred max|d| = {dp:.3e} loss {float(loss):.6f} d = {abs(float(oloss) - float(loss)):.3e}")
    assert dp < 1e-4 and abs(float(oloss) - float(loss)) < 1e-5
    for n in grad_names:
        dg = maxdiff(sdg[n].grad, grads[n])
        print(f"[{name}]   grad {n}: max|d| = {dg:.3e}  (|g|max {float(grads[n].abs().max()):.3e})")
        assert dg < 1e-4 * max(1.0, float(grads[n].abs().max()))
    torch.save(fx, os.path.join(OUT, name + ".pt"))


def gen_encoder_variants(name, seed):
    """Encoder (base, no interpolation -> ValueError) and OMREncoder with PE interpolation (M:290-332)."""
    torch.manual_seed(seed)
    enc = OMREncoder(4, 6, 10, num_layers=2, hidden_dim=32, num_heads=2, mlp_dim=64).eval()
    imgs = [torch.rand(1, 8, 16), torch.rand(1, 28, 44), torch.rand(1, 12, 48)]  # 2nd/3rd exceed the 6x10 grid
    with torch.no_grad():
        lat, mask = enc(imgs)
    sd = sd_cpu(enc)
    olat, omask = O.encoder_forward_padded(imgs, sd, "", 4, 2, "omr", "fp32", final_norm_bias_fill=sd["encoder_blocks.norm.bias"])
    assert torch.equal(mask, omask)
    d = maxdiff(olat, lat)
    print(f"[{name}] OMREncoder w/ interpolation max|d| = {d:.3e}")
    assert d < 2e-5
    base = Encoder(4, 6, 10, num_layers=1, hidden_dim=32, num_heads=2, mlp_dim=64)
    try:
        base([torch.rand(1, 28, 44)])
        raise AssertionError("expected ValueError")
    except ValueError as e:
        msg = str(e)
    try:
        O.encoder_forward([torch.rand(1, 28, 44)], sd_cpu(base), "", 4, 2, "base")
        raise AssertionError("expected ValueError")
    except ValueError as e:
        assert str(e) == msg
    torch.save(dict(state_dict=sd, imgs=imgs, latent=lat, mask=mask, too_large_msg=msg), os.path.join(OUT, name + ".pt"))


def gen_grpo(name, cfg, seed, lat_lens, group, top_k, temperature, max_actions):
    """GRPOViTOMR.cached_forward_rollout_policy (M:988-1049) under torch.manual_seed, fp32 and autocast(bf16): rollouts, log-probs, mask and
    the per-step logits.  torch.multinomial's stream cannot be reproduced elsewhere; what IS pinned: for the tokens the reference drew, the
    logits, the log-probs (log_softmax over the kept logits, M:1017-1018) and the mask.  `uniforms` are the inverse-CDF arguments that make
    the oracle's (and the HIP kernel's) draw land on the reference's token at every live step."""
    torch.manual_seed(seed)
    tf = build_vitomr(cfg)
    with torch.no_grad():
        for n, p in tf.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.add_(0.05 * torch.randn_like(p))
        tf.decoder.unembed.weight.mul_(4.0)
        tf.decoder.unembed.bias[2] += 5.0      # some rollouts reach <eos> early: exercises the ragged mask / clipping
    sd = sd_cpu(tf)
    E, B, S = cfg["dec_dim"], len(lat_lens), max(lat_lens)
    g = torch.Generator().manual_seed(seed + 1)
    mem = torch.randn(B, S, E, generator=g)                    # decoder-side memory (what the GRPO loop hands the policy)
    mask = torch.arange(S).unsqueeze(0) >= torch.tensor(lat_lens).unsqueeze(1)
    fx = dict(cfg=cfg, state_dict=sd, mem=mem, mask=mask, lat_lens=lat_lens, group=group, top_k=top_k, temperature=temperature,
              max_actions=max_actions)
    for tag, use_ac, cdt in (("fp32", False, torch.float), ("bf16", True, torch.bfloat16)):
        cached = tf.decoder.to_cached_version(B * group, cdt)
        cached.load_state_dict(tf.decoder.state_dict())
        grpo = GRPOViTOMR(tf.encoder, tf.transition_head, cached, sd).eval()
        mem_x, mask_x = grpo.expand_img_latent_for_rollout(mem, mask, group)
        rec = []
        orig = cached.cached_generate

        def wrapped(token_t, time_step, latent_attention_mask=None):
            out = orig(token_t, time_step, latent_attention_mask)
            rec.append(out.detach().float().squeeze(1).clone())
            return out

        cached.cached_generate = wrapped
        torch.manual_seed(seed + 2)
        with torch.no_grad():
            if use_ac:
                with autocast(device_type="cpu", dtype=torch.bfloat16):
                    ro, lp, mk = grpo.cached_forward_rollout_policy(mem_x, mask_x, max_actions=max_actions, top_k=top_k, temperature=temperature)
            else:
                ro, lp, mk = grpo.cached_forward_rollout_policy(mem_x, mask_x, max_actions=max_actions, top_k=top_k, temperature=temperature)
        logits = torch.stack(rec, 1)                               # (R, steps, V)
        R, T = ro.shape
        # uniforms that reproduce the reference's draws through the inverse-CDF restatement (dead positions: 0.5)
        u = torch.full((R, max_actions), 0.5)
        for t in range(1, T):
            ut, kept = O.rollout_uniform_for_token(logits[:, t - 1], ro[:, t].clamp(min=0), top_k, temperature)
            live = mk[:, t]
            assert bool(kept[live].all())
            u[live, t] = ut[live]
            tok, olp = O.rollout_sample_step(logits[:, t - 1], u[:, t], top_k, temperature, round_lp=use_ac)
            assert torch.equal(tok[live], ro[live, t]), (tag, t)
            # same logits -> same log-prob; under autocast aten's bf16 log_softmax kernel is not "fp32, rounded once": allow 2 bf16 ulps
            if use_ac:
                ulp = torch.exp2(torch.floor(torch.log2(lp[live, t].abs().clamp(min=2.0 ** -126))) - 7)
                assert bool(((olp[live] - lp[live, t]).abs() <= 2 * ulp).all()), (tag, t)
            else:
                assert maxdiff(olp[live], lp[live, t]) < 1e-5, (tag, t)
        fx[tag] = dict(rollouts=ro, log_probs=lp, mask=mk, step_logits=logits, uniforms=u)
        # the whole loop through the oracle (its own logits): fp32 must land on the reference's rollouts
        lens_x = [l for l in lat_lens for _ in range(group)]
        packed = torch.cat([mem[b, :l] for b, l in enumerate(lat_lens) for _ in range(group)], 0)
        prec = "bf16" if use_ac else "fp32"
        oro, olps, omk, ologits = O.rollout_generate(packed, lens_x, sd, cfg["dec_heads"], prec, max_actions, top_k, temperature, u, return_logits=True)
        # rows past their <eos> keep drawing junk that the returned rollouts no longer hold: logits are comparable on live positions only
        n = min(ologits.shape[1], logits.shape[1], T - 1)
        live_l = mk[:, 1:n + 1]
        dl = maxdiff(ologits[:, :n][live_l], logits[:, :n][live_l]) if torch.equal(oro, ro) else float("nan")
        print(f"[{name}] {tag}: rollouts {tuple(ro.shape)} lens {mk.sum(-1).tolist()}  oracle loop tokens equal = {torch.equal(oro, ro)}  logits max|d| = {dl:.3e}")
        if not use_ac:
            assert torch.equal(oro, ro) and torch.equal(omk, mk) and maxdiff(olps, lp) < 1e-4 and dl < 1e-4
        fx[f"oracle_{tag}_tokens_equal"] = bool(torch.equal(oro, ro))
    torch.save(fx, os.path.join(OUT, name + ".pt"))


def gen_ce_label_smoothing(name, seed):
    """OMRCELoss(pad_idx, label_smoothing) (M:784-796): loss and d loss / d logits for eps in {0, 0.1}."""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(4, 7, 227, generator=g) * 2.0
    target = torch.randint(0, 227, (4, 7), generator=g)
    target[0, 4:] = 1
    target[2, 1:] = 1
    fx = dict(logits=logits, target=target, pad_idx=1)
    for eps in (0.0, 0.1):
        lg = logits.clone().requires_grad_(True)
        loss = OMRCELoss(1, label_smoothing=eps)(lg, target)
        loss.backward()
        ol = O.ce_loss(logits, target, 1, label_smoothing=eps)
        assert abs(float(ol) - float(loss)) < 1e-6, (eps, float(ol), float(loss))
        fx[f"loss_{eps}"] = loss.detach()
        fx[f"grad_{eps}"] = lg.grad.detach().clone()
    print(f"[{name}] label smoothing: loss(0) = {float(fx['loss_0.0']):.6f} loss(0.1) = {float(fx['loss_0.1']):.6f}")
    torch.save(fx, os.path.join(OUT, name + ".pt"))


def main():
    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(1234)
    small = dict(P=4, pe_h=6, pe_w=10, ft_depth=1, enc_layers=2, enc_dim=32, enc_heads=2, enc_mlp=64, head_dim=64,
                 dec_layers=2, dec_dim=48, dec_heads=4, dec_mlp=96, max_len=24, gen_len=12, lmx_lens=[5, 9, 3])
    imgs_small = [torch.rand(1, 8, 16, generator=g), torch.rand(1, 24, 40, generator=g), torch.rand(1, 12, 20, generator=g)]
    gen_vitomr("vitomr_small", small, imgs_small, seed=11)
    gen_teacher_forced("tf_small", dict(small, ft_depth=2), imgs_small, seed=12)

    # real head dims (d_h = 64) and 16x16 patches at toy depth
    dh64 = dict(P=16, pe_h=4, pe_w=8, ft_depth=2, enc_layers=2, enc_dim=128, enc_heads=2, enc_mlp=256, head_dim=256,
                dec_layers=2, dec_dim=128, dec_heads=2, dec_mlp=256, max_len=24, gen_len=10, lmx_lens=[7, 4])
    imgs_dh64 = [torch.rand(1, 32, 64, generator=g), torch.rand(1, 64, 128, generator=g)]
    gen_vitomr("vitomr_dh64", dh64, imgs_dh64, seed=21)
    gen_teacher_forced("tf_dh64", dh64, imgs_dh64, seed=22)

    # odd head count (tests/test_vitomr.py:12 convention, num_heads=1): torch slow path even in eval
    odd = dict(small, enc_heads=1, dec_heads=1, enc_dim=10, enc_mlp=1, dec_dim=12, dec_mlp=5, head_dim=7, ft_depth=2)
    gen_vitomr("vitomr_odd", odd, imgs_small, seed=31)

    mae_cfg = dict(mask_ratio=0.75, P=4, pe_h=6, pe_w=10, enc_dim=32, dec_dim=16,
                   enc_kwargs=dict(num_layers=2, num_heads=2, mlp_dim=64), dec_kwargs=dict(num_layers=2, num_heads=2, mlp_dim=32))
    tg = [torch.rand(1, 8, 16, generator=g), torch.rand(1, 24, 40, generator=g), torch.rand(1, 12, 20, generator=g)]
    gen_mae("mae_small", mae_cfg, imgs_small, tg, seed=41)

    # the reference's own debug checkpoint (debug_pretrained_mae.pth: hidden 10, 2+2 layers, 1 head, mlp 1)
    dbg = torch.load("debug_pretrained_mae.pth")
    dbg_cfg = dict(mask_ratio=0.75, P=16, pe_h=60, pe_w=200, enc_dim=10, dec_dim=10,
                   enc_kwargs=dict(num_layers=2, num_heads=1, mlp_dim=1), dec_kwargs=dict(num_layers=2, num_heads=1, mlp_dim=1))
    dimgs = [torch.rand(1, 32, 64, generator=g), torch.rand(1, 48, 32, generator=g)]
    # keep the fixture small: store only the PE rows these images touch
    gen_mae("mae_debug_ckpt", dbg_cfg, dimgs, dimgs, seed=51, sd_override=dbg)

    gen_encoder_variants("omr_encoder_interp", seed=61)

    # ---- round 2 additions (own generators: the fixtures above stay bit-identical) ----------------------------------------
    # GRPO rollout policy (M:988-1049): 2 memories x 3 rollouts on the small decoder (d_h = 12), 2 x 4 at d_h = 64
    gen_grpo("grpo_small", small, seed=71, lat_lens=[9, 5], group=3, top_k=5, temperature=1.2, max_actions=10)
    gen_grpo("grpo_dh64", dh64, seed=72, lat_lens=[24, 40], group=4, top_k=8, temperature=0.9, max_actions=9)
    # teacher-forced TRAIN step whose images exceed the PE grid: batchify interpolates in every mode (M:304-332) and the
    # gradient reaches pos_embedding through the bilinear interpolation (full fine-tune: pos_embedding trainable, M:667-677)
    g2 = torch.Generator().manual_seed(4321)
    interp = dict(small, ft_depth=2, lmx_lens=[5, 9, 3])
    imgs_interp = [torch.rand(1, 8, 16, generator=g2), torch.rand(1, 28, 44, generator=g2), torch.rand(1, 12, 48, generator=g2)]
    gen_teacher_forced("tf_interp", interp, imgs_interp, seed=81)
    gen_ce_label_smoothing("ce_label_smoothing", seed=91)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
