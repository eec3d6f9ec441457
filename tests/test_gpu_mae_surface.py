"""The three MAE helper methods the reference pins with exact-tensor KATs (/root/reference/tests/test_mae.py:36-55, 57-79, 89-122), restated
on the mirror (GPU; identity projection, labelled positional grids), plus the imported reference's own outputs of the same methods on the
`mae_small` weights (tests/golden/mae_surface.pt, written by oracle/gen_surface.py)."""
import pytest
import torch
from torch import nn

from conftest import load_golden

pytestmark = pytest.mark.gpu

PE_MAX_HEIGHT, PE_MAX_WIDTH = 60, 200     # acai_omr/train/pre_train.py:21-22
NUM_CHANNELS = 1


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from acai_omr_amd import _lib
    _lib.lib()
    return "cuda"


def md(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def test_kat_mask_sequence(dev):
    """tests/test_mae.py:36-55: labelled patches, labelled PE grid with -1 filler; shapes, lengths, and the shuffle is undone by ids_restore."""
    from acai_omr_amd.models.models import MAEEncoder
    encoder = MAEEncoder(0.50, 2, PE_MAX_HEIGHT, PE_MAX_WIDTH, num_heads=1, hidden_dim=1).to(dev)
    SEQ_LEN = 4
    x = torch.arange(SEQ_LEN).unsqueeze(0).repeat(12, 1).unsqueeze(0).to(dev)      # (1, 12, 4), patch i holds the value i
    pe_num_grid = torch.arange(4, dtype=torch.float).reshape(2, 2)
    pe_filler = torch.zeros(2, 4) - 1
    encoder.pos_embedding = nn.Parameter(torch.cat((pe_num_grid, pe_filler), dim=1).unsqueeze(-1).to(dev))
    t_masked, pos_embed_slice, unmasked_seq_len, len_keep, seq_mask, ids_restore = encoder.mask_sequence(x, 2, 2)
    assert t_masked.shape == torch.Size([1, 12, 2])
    assert unmasked_seq_len == SEQ_LEN
    assert len_keep == 2
    assert seq_mask.dtype == torch.int32 and int(seq_mask.sum()) == SEQ_LEN - len_keep
    # the PE rows follow the kept patches: patch label i <-> PE label i, and the -1 filler never shows
    assert torch.equal(pos_embed_slice.reshape(-1).cpu(), t_masked[0, 0].float().cpu())
    t_full = torch.concat((t_masked, (torch.zeros(1, 12, 3, device=dev, dtype=t_masked.dtype) - 1)), dim=-1)
    undo = t_full.index_select(dim=-1, index=ids_restore.squeeze(0))
    assert undo.shape[:2] == x.shape[:2]
    # kept patches return to their original positions, masked positions show the -1 mask token
    u = undo[0, 0, :SEQ_LEN].cpu()
    for pos in range(SEQ_LEN):
        assert int(u[pos]) == (pos if int(seq_mask[pos]) == 0 else -1)
    # injected noise decides the permutation: argsort([.3,.1,.9,.2]) = [1,3,0,2] -> keep patches 1 and 3
    out = encoder.mask_sequence(x, 2, 2, noise=torch.tensor([0.3, 0.1, 0.9, 0.2], device=dev))
    assert out[0][0, 0].tolist() == [1, 3] and out[4].tolist() == [1, 0, 1, 0] and out[5].tolist() == [2, 0, 3, 1]
    assert out[1].reshape(-1).tolist() == [1.0, 3.0]


def test_kat_masked_encoder_batchify(dev):
    """tests/test_mae.py:57-79: identity projection + all-ones PE -> kept rows are 2, padded rows 0; both attention masks."""
    from acai_omr_amd.models.models import MAEEncoder
    patch_size = 2
    hidden_dim = NUM_CHANNELS * patch_size ** 2
    encoder = MAEEncoder(0.50, patch_size, PE_MAX_HEIGHT, PE_MAX_WIDTH, hidden_dim=hidden_dim, num_heads=1).to(dev)
    encoder.projection = nn.Identity()
    encoder.pos_embedding = nn.Parameter(torch.ones(50, 50, hidden_dim, device=dev))
    x = [torch.ones(NUM_CHANNELS, 4, 4), torch.ones(NUM_CHANNELS, 4, 6)]
    out = encoder.batchify(x)
    assert len(out) == 8
    embeddings, encoder_attn_mask, decoder_attn_mask, kept_seq_lens, unmasked_seq_lens, seq_masks, ids_restores, patchified_dims = out
    first = torch.cat([torch.ones(NUM_CHANNELS, 2, hidden_dim) + 1, torch.zeros(NUM_CHANNELS, 1, hidden_dim)], dim=1)
    second = torch.ones(NUM_CHANNELS, 3, hidden_dim) + 1
    assert torch.equal(torch.cat([first, second]), embeddings.cpu())
    assert torch.equal(encoder_attn_mask.cpu(), torch.cat(((torch.arange(3) >= 2).unsqueeze(0), (torch.arange(3) >= 3).unsqueeze(0))))
    assert torch.equal(decoder_attn_mask.cpu(), torch.cat(((torch.arange(6) >= 4).unsqueeze(0), (torch.arange(6) >= 6).unsqueeze(0))))
    assert kept_seq_lens == [2, 3] and unmasked_seq_lens == [4, 6] and patchified_dims == [(2, 2), (2, 3)]
    assert seq_masks.is_nested and ids_restores.is_nested
    assert [int(m.sum()) for m in seq_masks.unbind()] == [2, 3]
    assert [sorted(r.tolist()) for r in ids_restores.unbind()] == [list(range(4)), list(range(6))]
    # MAEEncoder.forward keeps its 7-tuple (tests/test_mae.py:81-87)
    # (the reference's test uses hidden 200 over 2 heads; the attention kernels carry head dims up to 64, so 4 heads here)
    enc2 = MAEEncoder(0.50, 2, PE_MAX_HEIGHT, PE_MAX_WIDTH, num_layers=2, num_heads=4, hidden_dim=200, mlp_dim=500).to(dev)
    f = enc2([torch.rand(NUM_CHANNELS, 4, 4), torch.rand(NUM_CHANNELS, 4, 8)])
    assert len(f) == 7 and f[0].shape == torch.Size([2, 4, 200])


def test_kat_prepare_for_decoder(dev):
    """tests/test_mae.py:89-122, value for value: mask tokens of 100 in front, labelled latents in ascending order, +500 PE, zero padding."""
    from acai_omr_amd.models.models import MAE
    mae = MAE(0.5, 1, PE_MAX_HEIGHT, PE_MAX_WIDTH, encoder_hidden_dim=2, decoder_hidden_dim=1, encoder_kwargs={"num_heads": 1},
              decoder_kwargs={"num_heads": 1}).to(dev)
    mae.mask_token = nn.Parameter(torch.zeros(1, 1, 1, device=dev) + 100)
    first_latent_seq = torch.cat([(torch.arange(2) + 1).unsqueeze(-1).unsqueeze(0), torch.zeros(1, 1, 1) - 1], dim=1)
    second_latent_seq = (torch.arange(3) + 1).unsqueeze(-1).unsqueeze(0)
    first_latent_seq = first_latent_seq.index_select(dim=1, index=torch.tensor([1, 0, 2]))
    second_latent_seq = second_latent_seq.index_select(dim=1, index=torch.tensor([2, 0, 1]))
    kept_seq_lens = [2, 3]
    unmasked_seq_lens = [4, 6]
    patchified_dims = [(2, 2), (2, 3)]
    batch_ids_restore = torch.nested.nested_tensor([torch.tensor([2, 3, 1, 0]), torch.tensor([3, 4, 5, 1, 2, 0])], layout=torch.jagged)
    latent = torch.cat([first_latent_seq, second_latent_seq])
    pe_num_grid = torch.zeros(2, 3) + 500
    pe_filler = torch.zeros(2, 4) - 1
    mae.decoder_pos_embedding = nn.Parameter(torch.cat((pe_num_grid, pe_filler), dim=1).unsqueeze(-1).to(dev))
    reconstructed_seq = mae.prepare_for_decoder(latent.to(dev), kept_seq_lens, unmasked_seq_lens, batch_ids_restore, patchified_dims)
    first_expected_seq = torch.cat([torch.tensor([100, 100, 1, 2]).unsqueeze(-1).unsqueeze(0) + 500,
                                    torch.tensor([0, 0]).unsqueeze(-1).unsqueeze(0)], dim=1)
    second_expected_seq = torch.tensor([100, 100, 100, 1, 2, 3]).unsqueeze(-1).unsqueeze(0) + 500
    assert torch.equal(reconstructed_seq.cpu(), torch.cat([first_expected_seq, second_expected_seq]).float())


def _mae_small(dev):
    from acai_omr_amd.models.models import MAE
    fx = load_golden("mae_small")
    cfg = fx["cfg"]
    mae = MAE(cfg["mask_ratio"], cfg["P"], cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"], decoder_hidden_dim=cfg["dec_dim"],
              encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"])
    mae.load_state_dict(fx["state_dict"])
    return mae.to(dev).eval(), fx


def test_mae_helpers_vs_reference_outputs(dev):
    """mae_surface.pt: the imported reference's batchify / mask_sequence / prepare_for_decoder on real (seeded) weights and injected noise."""
    mae, fx = _mae_small(dev)
    ref = load_golden("mae_surface")
    imgs, noises = fx["imgs"], fx["noises"]
    with torch.no_grad():
        b = mae.encoder.batchify(list(imgs), noises=noises)
    rb = ref["batchify"]
    assert b[0].shape == rb["embeddings"].shape and md(b[0], rb["embeddings"]) < 1e-5      # padded rows hold the projection bias
    assert torch.equal(b[1].cpu(), rb["encoder_attention_mask"]) and torch.equal(b[2].cpu(), rb["decoder_attention_mask"])
    assert b[3] == rb["kept_seq_lens"] and b[4] == rb["unmasked_seq_lens"] and [tuple(d) for d in b[7]] == [tuple(d) for d in rb["patchified_dims"]]
    for mine, theirs in zip(b[5].unbind(), rb["seq_masks"]):
        assert mine.dtype == theirs.dtype and torch.equal(mine.cpu(), theirs)
    for mine, theirs in zip(b[6].unbind(), rb["ids_restore"]):
        assert torch.equal(mine.cpu(), theirs)

    rm = ref["mask_sequence"]
    i, P = rm["image"], fx["cfg"]["P"]
    t = torch.nn.functional.unfold(imgs[i].unsqueeze(0), kernel_size=P, stride=P).to(dev)
    with torch.no_grad():
        ms = mae.encoder.mask_sequence(t, imgs[i].shape[-2] // P, imgs[i].shape[-1] // P, noise=noises[i].to(dev))
    assert torch.equal(ms[0].cpu(), rm["t_masked"]) and md(ms[1], rm["pos_embed_slice"]) == 0.0
    assert (ms[2], ms[3]) == (rm["unmasked_seq_len"], rm["len_keep"])
    assert torch.equal(ms[4].cpu(), rm["seq_mask"]) and torch.equal(ms[5].cpu(), rm["ids_restore"])

    rp = ref["prepare_for_decoder"]
    restore = torch.nested.as_nested_tensor([r.to(dev) for r in rp["ids_restore"]], layout=torch.jagged)
    with torch.no_grad():
        out = mae.prepare_for_decoder(rp["latent"].to(dev), rp["kept_seq_lens"], rp["unmasked_seq_lens"], restore, rp["patchified_dims"])
    assert out.shape == rp["out"].shape and md(out, rp["out"]) < 1e-6

    # and the chain MAE.forward runs (encoder -> decoder_embed -> prepare_for_decoder) reproduces the reference's decoder input
    with torch.no_grad():
        lat, dmask, kept, lens, smasks, rest, dims = mae.encoder(list(imgs), noises=noises)
        valid = ~ref["encoder_forward"]["decoder_attention_mask"]
        assert torch.equal(dmask.cpu(), ref["encoder_forward"]["decoder_attention_mask"])
        kmask = torch.arange(lat.shape[1]).unsqueeze(0) < torch.tensor(kept).unsqueeze(1)
        assert md(lat.cpu()[kmask], ref["encoder_forward"]["latent"][kmask]) < 1e-4
        lat_d = torch.nn.functional.linear(lat, mae.decoder_embed.weight, mae.decoder_embed.bias)
        out2 = mae.prepare_for_decoder(lat_d, kept, lens, rest, dims)
    assert md(out2.cpu()[valid], rp["out"][valid]) < 1e-4


def test_prepare_for_decoder_gradients(dev):
    """prepare_for_decoder is differentiable in the latent, the mask token and the decoder PE (MAE.forward trains through it)."""
    mae, fx = _mae_small(dev)
    ref = load_golden("mae_surface")["prepare_for_decoder"]
    lat = ref["latent"].to(dev).requires_grad_(True)
    restore = torch.nested.as_nested_tensor([r.to(dev) for r in ref["ids_restore"]], layout=torch.jagged)
    out = mae.prepare_for_decoder(lat, ref["kept_seq_lens"], ref["unmasked_seq_lens"], restore, ref["patchified_dims"])
    w = torch.arange(out.numel(), device=dev, dtype=torch.float32).reshape(out.shape) / out.numel()
    (out * w).sum().backward()
    # CPU restatement with torch ops (models.py:219-241)
    latc = ref["latent"].clone().requires_grad_(True)
    mt = mae.mask_token.detach().cpu().clone().requires_grad_(True)
    dpe = mae.decoder_pos_embedding.detach().cpu().clone().requires_grad_(True)
    seqs = []
    for i, (k, n) in enumerate(zip(ref["kept_seq_lens"], ref["unmasked_seq_lens"])):
        s = torch.cat([latc[i, :k], mt.reshape(1, -1).expand(n - k, -1)], 0)[ref["ids_restore"][i]]
        h, wd = ref["patchified_dims"][i]
        s = s + dpe[:h, :wd].reshape(-1, dpe.shape[-1])
        seqs.append(torch.cat([s, torch.zeros(out.shape[1] - n, s.shape[1])], 0))
    (torch.stack(seqs) * w.cpu()).sum().backward()
    assert md(lat.grad, latc.grad) < 1e-6
    assert md(mae.mask_token.grad, mt.grad) < 1e-5
    assert md(mae.decoder_pos_embedding.grad, dpe.grad) < 1e-6


def test_uncached_rollout_policy(dev):
    """GRPOViTOMR.uncached_forward_rollout_policy (deprecated upstream, models.py:897-945): shapes, mask and log-prob contract; with top_k = 1
    it is the greedy decode, whatever the multinomial draws."""
    from conftest import VOCAB
    from acai_omr_amd.models.models import FineTuneOMREncoder, GRPOViTOMR, OMRDecoder, TeacherForcedViTOMR
    fx = load_golden("vitomr_small")
    cfg, sd = fx["cfg"], fx["state_dict"]

    enc = FineTuneOMREncoder(cfg["P"], cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                             num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"])
    dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"])
    tf = TeacherForcedViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"])
    tf.load_state_dict(sd)
    g = GRPOViTOMR(tf.encoder, tf.transition_head, dec, sd).to(dev).eval()
    with torch.no_grad():
        lat, mask = g.encoder(fx["imgs"])
        mem = g.transition_head(lat)
        r, lp, m = g.uncached_forward_rollout_policy(mem, mask, max_actions=cfg["gen_len"], top_k=1, temperature=1.2)
    assert r.shape == (len(fx["imgs"]), cfg["gen_len"]) and lp.shape == r.shape and m.dtype == torch.bool
    # top_k = 1 makes every draw the argmax of the UNCACHED forward on the prefix (positions 0..t-1: not the cached loop's off-by-one)
    with torch.no_grad():
        seqs = torch.full_like(r, g.decoder.pad_idx)
        seqs[:, 0] = g.decoder.bos_idx
        for t in range(1, cfg["gen_len"]):
            seqs[:, t] = g.decoder.generate(seqs[:, :t], mem, latent_attention_mask=mask)[:, -1, :].float().argmax(-1)
            if bool((seqs == g.decoder.eos_idx).any(-1).all()):
                break
    assert torch.equal(r.masked_fill(~m, g.decoder.pad_idx), seqs.masked_fill(~g.create_inference_mask(seqs), g.decoder.pad_idx))
    assert float(lp.abs().max()) == 0.0          # one kept logit: log_softmax over the masked vocabulary is 0 for the survivor
    assert torch.equal(r[~m].cpu(), torch.full_like(r[~m].cpu(), g.decoder.pad_idx))
