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

from acai_omr.models.models import (MAE, FineTuneOMREncoder, MAELoss, OMRCELoss, OMRDecoder, OMREncoder,  # noqa: E402
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
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
