"""Oracle (oracle/vitomr_oracle.py) pinned against the committed golden vectors, which were
produced by the imported reference (oracle/gen_golden.py), and against the reference's own
known-answer tests.  CPU only."""
import pytest
import torch

from conftest import load_golden
from oracle import vitomr_oracle as O


def md(a, b):
    return float((a.float() - b.float()).abs().max())


@pytest.mark.parametrize("name", ["vitomr_small", "vitomr_dh64", "vitomr_dh64b", "vitomr_odd"])
def test_encoder_head_decode_fp32(name):
    fx = load_golden(name)
    cfg, sd, ref = fx["cfg"], fx["state_dict"], fx["ref_fp32"]
    fill = sd["encoder.fine_tune_blocks.norm.bias"] if cfg["enc_heads"] % 2 == 0 else None
    lat, mask = O.encoder_forward_padded(fx["imgs"], sd, "encoder.", cfg["P"], cfg["enc_heads"], "omr_ft", "fp32", fill)
    assert torch.equal(mask, ref["latent_mask"])
    assert md(lat[~mask], ref["latent"][~mask]) < 2e-5
    if fill is not None:  # eval fast path: padded rows equal the final norm's bias
        assert md(lat, ref["latent"]) < 2e-5
    packed, lens = O.unpad(lat, mask)
    mem = O.transition_head(packed, sd, "fp32")
    assert md(O.pad_packed(mem, lens)[0][~mask], ref["memory"][~mask]) < 2e-5
    seqs, lps, smask, logits = O.greedy_generate(mem, lens, sd, cfg["dec_heads"], "fp32", cfg["gen_len"], return_logits=True)
    assert torch.equal(seqs, ref["seqs"])           # bit-exact token ids
    assert torch.equal(smask, ref["seq_mask"])
    assert md(lps, ref["log_probs"]) < 1e-4
    assert md(logits[:, :ref["step_logits"].shape[1]], ref["step_logits"]) < 1e-4  # well inside the 1e-3 bar


@pytest.mark.parametrize("name", ["vitomr_small", "vitomr_dh64", "vitomr_dh64b", "vitomr_odd"])
def test_decode_bf16_autocast_plumbing(name):
    """inference() plumbing: fp32 encoder, autocast(bf16) head + decode with a bf16 KV cache."""
    fx = load_golden(name)
    cfg, sd, ref = fx["cfg"], fx["state_dict"], fx["ref_bf16"]
    seqs, lps, smask = O.vitomr_inference(fx["imgs"], sd, cfg["enc_heads"], cfg["dec_heads"], cfg["P"], cfg["gen_len"])
    assert torch.equal(seqs, ref["seqs"])
    assert torch.equal(smask, ref["seq_mask"])
    assert md(lps, ref["log_probs"]) < 0.13  # bf16 log-probs: one or two ulps at |x| ~ 4-8


@pytest.mark.parametrize("name", ["tf_small", "tf_dh64", "tf_interp"])
def test_teacher_forced_loss_and_grads(name):
    """tf_interp: two of the three images exceed the PE grid, so batchify interpolates IN THE TRAINING PATH (models.py:304-332) and the
    gradient of pos_embedding flows through the bilinear interpolation."""
    fx = load_golden(name)
    cfg = fx["cfg"]
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx["state_dict"].items()}
    pred, tgt = O.teacher_forced_forward(list(zip(fx["imgs"], fx["lmx"])), sd, cfg["enc_heads"], cfg["dec_heads"], cfg["P"], "fp32")
    assert torch.equal(tgt, fx["target"])
    valid = tgt != 1
    assert md(pred[valid], fx["pred"][valid]) < 1e-4
    loss = O.ce_loss(pred, tgt)
    assert abs(float(loss) - float(fx["loss"])) < 1e-5
    loss.backward()
    for n, g in fx["grads"].items():
        assert md(sd[n].grad, g) < 1e-4 * max(1.0, float(g.abs().max())), n


@pytest.mark.parametrize("name", ["mae_small", "mae_debug_ckpt"])
def test_mae_forward_loss_grads(name):
    fx = load_golden(name)
    cfg = fx["cfg"]
    sd = {k: v.clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
    batch = list(zip(fx["imgs"], fx["tgts"]))
    pred, lm, tgt, lens = O.mae_forward(batch, fx["noises"], sd, cfg["P"], cfg["mask_ratio"],
                                        cfg["enc_kwargs"]["num_heads"], cfg["dec_kwargs"]["num_heads"])
    ppred, pmask = O.pad_packed(pred, lens)
    assert torch.equal(O.pad_packed(lm, lens, False)[0], fx["loss_mask"])
    assert torch.equal(O.pad_packed(tgt, lens)[0], fx["target"])
    assert md(ppred[~pmask], fx["pred"][~pmask]) < 1e-4
    loss = O.mae_loss(pred, lm, tgt)
    assert abs(float(loss) - float(fx["loss"])) < 1e-5
    loss.backward()
    for n, g in fx["grads"].items():
        assert md(sd[n].grad, g) < 1e-4 * max(1.0, float(g.abs().max())), n


def test_omr_encoder_pe_interpolation_and_too_large_error():
    fx = load_golden("omr_encoder_interp")
    sd = fx["state_dict"]
    lat, mask = O.encoder_forward_padded(fx["imgs"], sd, "", 4, 2, "omr", "fp32", sd["encoder_blocks.norm.bias"])
    assert torch.equal(mask, fx["mask"])
    assert md(lat, fx["latent"]) < 2e-5
    with pytest.raises(ValueError) as e:
        O.encoder_forward([torch.rand(1, 28, 44)], sd, "", 4, 2, "base")
    assert str(e.value) == fx["too_large_msg"]


@pytest.mark.parametrize("name", ["grpo_small", "grpo_dh64"])
@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_grpo_rollout_policy_vs_reference(name, tag):
    """GRPOViTOMR.cached_forward_rollout_policy (models.py:988-1049) as the imported reference ran it under torch.manual_seed: with the
    fixture's uniforms (the inverse-CDF arguments of the reference's own draws) the oracle's sampling step reproduces every live token and
    its log-prob from the REFERENCE's per-step logits, and the oracle's whole rollout loop (its own logits) lands on the same rollouts."""
    fx = load_golden(name)
    cfg, ref = fx["cfg"], fx[tag]
    ro, lp, mk, logits, u = ref["rollouts"], ref["log_probs"], ref["mask"], ref["step_logits"], ref["uniforms"]
    bf = tag == "bf16"
    assert torch.equal(mk, O.inference_mask(ro, 2)) and bool((ro[~mk] == 1).all()) and bool((lp[~mk] == 0).all())
    assert mk.sum(-1).min() < ro.shape[1]            # ragged: some rollouts ended early
    for t in range(1, ro.shape[1]):
        tok, olp = O.rollout_sample_step(logits[:, t - 1], u[:, t], fx["top_k"], fx["temperature"], round_lp=bf)
        live = mk[:, t]
        assert torch.equal(tok[live], ro[live, t])
        if bf:   # aten's bf16 log_softmax kernel is not "fp32, rounded once": two bf16 ulps
            ulp = torch.exp2(torch.floor(torch.log2(lp[live, t].abs().clamp(min=2.0 ** -126))) - 7)
            assert bool(((olp[live] - lp[live, t]).abs() <= 2 * ulp).all())
        else:
            assert md(olp[live], lp[live, t]) < 1e-5
    G, lens = fx["group"], fx["lat_lens"]
    lens_x = [l for l in lens for _ in range(G)]
    packed = torch.cat([fx["mem"][b, :l] for b, l in enumerate(lens) for _ in range(G)], 0)
    oro, olps, omk = O.rollout_generate(packed, lens_x, fx["state_dict"], cfg["dec_heads"], tag, fx["max_actions"], fx["top_k"], fx["temperature"], u)
    assert fx[f"oracle_{tag}_tokens_equal"]
    assert torch.equal(oro, ro) and torch.equal(omk, mk)
    assert md(olps, lp) < (0.07 if bf else 1e-4)


def test_ce_loss_label_smoothing_vs_reference():
    """OMRCELoss(pad_idx, label_smoothing) (models.py:784-796) for eps = 0 and 0.1: value and gradient."""
    fx = load_golden("ce_label_smoothing")
    for eps in (0.0, 0.1):
        lg = fx["logits"].clone().requires_grad_(True)
        loss = O.ce_loss(lg, fx["target"], fx["pad_idx"], label_smoothing=eps)
        assert abs(float(loss) - float(fx[f"loss_{eps}"])) < 1e-6
        loss.backward()
        assert md(lg.grad, fx[f"grad_{eps}"]) < 1e-7


# ---- the reference's own known-answer tests, restated on the oracle ----------------------------
def test_kat_mae_loss():
    """tests/test_mae.py:169-180 -> 10.583329200744629"""
    target = torch.cat([torch.tensor([[1, 1, 1], [2, 2, 2]], dtype=torch.float).unsqueeze(-1).repeat(1, 1, 6),
                        torch.tensor([[2, 2, 2], [3, 3, 3]], dtype=torch.float).unsqueeze(-1).repeat(1, 1, 6)], dim=-1)
    pred = torch.tensor([[2, 2, 2], [3, 3, 4]], dtype=torch.float).unsqueeze(-1).repeat(1, 1, 12)
    loss_mask = torch.tensor([[1, 0, 0], [1, 0, 1]], dtype=torch.float)
    assert float(O.mae_loss(pred, loss_mask, target)) == 10.583329200744629


def test_kat_batchify_and_split():
    """tests/test_vitomr.py:151-172"""
    seqs = [torch.tensor([0, 2, 3, 226]), torch.tensor([0, 2, 2, 3, 4, 226])]
    inp, tgt, mask = O.batchify_and_split_lmx_seqs(seqs, 1)
    assert inp.tolist() == [[0, 2, 3, 226, 1], [0, 2, 2, 3, 4]]
    assert tgt.tolist() == [[2, 3, 226, 1, 1], [2, 2, 3, 4, 226]]
    assert mask.int().tolist() == [[0, 0, 0, 0, 1], [0, 0, 0, 0, 0]]


def test_kat_inference_mask():
    """tests/test_vitomr.py:415-436: [[bos,eos,10,eos],[bos,20,20,eos]] -> [[1,1,0,0],[1,1,1,1]]"""
    seqs = torch.tensor([[0, 2, 10, 2], [0, 20, 20, 2]])
    assert O.inference_mask(seqs, 2).int().tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]


def test_kat_encoder_batchify_padding():
    """tests/test_mae.py:8-24: identity projection, all-ones PE -> rows of 2, mask arange >= len."""
    P, E = 2, 4
    sd = {"pos_embedding": torch.ones(50, 50, E), "projection.weight": torch.eye(E), "projection.bias": torch.zeros(E)}
    x, lens = O.encoder_embed([torch.ones(1, 4, 4), torch.ones(1, 4, 8)], sd, "", P, False, "fp32")
    out, mask = O.pad_packed(x, lens)
    exp = torch.cat([torch.cat([torch.ones(1, 4, E) + 1, torch.zeros(1, 4, E)], 1), torch.ones(1, 8, E) + 1])
    assert torch.equal(out, exp)
    assert torch.equal(mask, torch.stack([torch.arange(8) >= 4, torch.arange(8) >= 8]))


def test_kat_mask_ids_q5():
    """Q5: len_keep = int(N * (1 - ratio)) (floor); seq_mask int32 with 1 = masked, in original order."""
    noise = torch.tensor([0.9, 0.1, 0.5, 0.3, 0.7])
    keep, restore, seq_mask, k = O.mae_mask_ids(noise, 0.75)
    assert k == 1 and keep.tolist() == [1]
    assert seq_mask.dtype == torch.int32 and seq_mask.tolist() == [1, 0, 1, 1, 1]
    assert restore.tolist() == [4, 0, 2, 1, 3]
