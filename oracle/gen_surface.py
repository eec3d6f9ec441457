"""Pin the drop-in Python surface and the three MAE helper methods the reference's own tests pin.

Run in the build container only (needs /root/reference, read-only):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python /root/repo/oracle/gen_surface.py

Writes
  tests/golden/surface.json   - public classes / methods / functions of the reference's models.py, kv_caching.py and
                                inference/vitomr_inference.py with parameter names, kinds and default values (as source text).  It is
                                read from the reference's files with `ast` (vitomr_inference.py cannot be imported here: torchvision is
                                not installed), so it is data ABOUT the interface - names and defaults - not source.
                                tests/test_surface.py diffs the mirror package against it.
  tests/golden/mae_surface.pt - outputs of the imported reference's MAEEncoder.mask_sequence / MAEEncoder.batchify /
                                MAE.prepare_for_decoder on the `mae_small` fixture's weights, images and noise (test infrastructure).
"""
import ast
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"
FILES = {
    "models.models": "acai_omr/models/models.py",
    "models.kv_caching": "acai_omr/models/kv_caching.py",
    "inference.vitomr_inference": "acai_omr/inference/vitomr_inference.py",
}


def params_of(fn):
    a = fn.args
    out = []
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for arg, d in zip(pos, defaults):
        out.append([arg.arg, None if d is None else ast.unparse(d), "pos"])
    if a.vararg:
        out.append([a.vararg.arg, None, "var"])
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append([arg.arg, None if d is None else ast.unparse(d), "kw"])
    if a.kwarg:
        out.append([a.kwarg.arg, None, "varkw"])
    return out


def surface_of(path):
    tree = ast.parse(open(path).read())
    mod = {"functions": {}, "classes": {}, "constants": []}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            mod["functions"][node.name] = params_of(node)
        elif isinstance(node, ast.ClassDef):
            fns = [n for n in node.body if isinstance(n, ast.FunctionDef)]
            is_prop = lambda n: any(ast.unparse(d) == "property" for d in n.decorator_list)
            methods = {n.name: params_of(n) for n in fns if not is_prop(n)}
            mod["classes"][node.name] = {"bases": [ast.unparse(b) for b in node.bases], "methods": methods,
                                         "properties": [n.name for n in fns if is_prop(n)]}
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                if isinstance(t, ast.Name) and t.id.isupper():
                    mod["constants"].append(t.id)
    return mod


def gen_surface_json():
    surf = {name: surface_of(os.path.join(REF, rel)) for name, rel in FILES.items()}
    with open(os.path.join(OUT, "surface.json"), "w") as f:
        json.dump(surf, f, indent=1, sort_keys=True)
    n = sum(len(c["methods"]) for m in surf.values() for c in m["classes"].values()) + sum(len(m["functions"]) for m in surf.values())
    print(f"surface.json: {n} callables over {len(surf)} modules")


def gen_mae_surface():
    from acai_omr.models.models import MAE
    fx = torch.load(os.path.join(OUT, "mae_small.pt"), map_location="cpu", weights_only=False)
    cfg = fx["cfg"]
    mae = MAE(cfg["mask_ratio"], cfg["P"], cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"], decoder_hidden_dim=cfg["dec_dim"],
              encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"])
    mae.load_state_dict(fx["state_dict"])
    mae.eval()
    imgs, noises = fx["imgs"], fx["noises"]
    real_rand = torch.rand
    queue = []

    def fake_rand(n, device=None):          # the reference draws torch.rand(L, device=...) once per image, in order (models.py:110)
        x = queue.pop(0)
        assert x.numel() == n
        return x.clone()

    out = {}
    with torch.no_grad():
        torch.rand = fake_rand
        try:
            queue[:] = list(noises)
            b = mae.encoder.batchify(list(imgs))
            out["batchify"] = dict(embeddings=b[0], encoder_attention_mask=b[1], decoder_attention_mask=b[2], kept_seq_lens=b[3],
                                   unmasked_seq_lens=b[4], seq_masks=[t.clone() for t in b[5].unbind()],
                                   ids_restore=[t.clone() for t in b[6].unbind()], patchified_dims=b[7])
            # mask_sequence on the second image's unfolded patches
            i = 1
            P = cfg["P"]
            h_p, w_p = imgs[i].shape[-2] // P, imgs[i].shape[-1] // P
            t = mae.encoder.unfold(imgs[i].unsqueeze(0))
            queue[:] = [noises[i]]
            ms = mae.encoder.mask_sequence(t, h_p, w_p)
            out["mask_sequence"] = dict(image=i, t_masked=ms[0], pos_embed_slice=ms[1], unmasked_seq_len=ms[2], len_keep=ms[3], seq_mask=ms[4],
                                        ids_restore=ms[5])
            # MAEEncoder.forward -> decoder_embed -> prepare_for_decoder, as MAE.forward chains them (models.py:254-257)
            queue[:] = list(noises)
            lat, dmask, kept, lens, smasks, restore, dims = mae.encoder(list(imgs))
            lat_d = mae.decoder_embed(lat)
            prep = mae.prepare_for_decoder(lat_d, kept, lens, restore, dims)
            out["prepare_for_decoder"] = dict(latent=lat_d, kept_seq_lens=kept, unmasked_seq_lens=lens,
                                              ids_restore=[t.clone() for t in restore.unbind()], patchified_dims=dims, out=prep)
            out["encoder_forward"] = dict(latent=lat, decoder_attention_mask=dmask)
        finally:
            torch.rand = real_rand
    torch.save(out, os.path.join(OUT, "mae_surface.pt"))
    print("mae_surface.pt: batchify", tuple(out["batchify"]["embeddings"].shape), "prepare_for_decoder", tuple(out["prepare_for_decoder"]["out"].shape))


def gen_distinct_rows(name="vitomr_dh64b"):
    """A d_h = 64 ViTOMR fixture whose two images decode to DIFFERENT token rows (round-2 verdict: in `vitomr_dh64` both rows are the same ten
    tokens - the scaled unembed dominates the logits there - so a token check barely sees the cross-attention).  Same generator as the
    other ViTOMR fixtures (oracle/gen_golden.py: gen_vitomr, incl. its oracle == reference asserts) with the decoder's cross-attention output
    projections scaled up at construction; seeds are tried until the rows differ in at least half of their positions and the oracle's
    autocast restatement lands on the reference's bf16 tokens."""
    import gen_golden as G
    dh64 = dict(P=16, pe_h=4, pe_w=8, ft_depth=2, enc_layers=2, enc_dim=128, enc_heads=2, enc_mlp=256, head_dim=256,
                dec_layers=2, dec_dim=128, dec_heads=2, dec_mlp=256, max_len=24, gen_len=12, lmx_lens=[7, 4])
    g = torch.Generator().manual_seed(777)
    imgs = [torch.rand(1, 32, 64, generator=g), torch.rand(1, 64, 128, generator=g), torch.rand(1, 48, 96, generator=g)]
    plain_build = G.build_vitomr

    def build(cfg, dropout_zero=False):
        m = plain_build(cfg, dropout_zero)
        with torch.no_grad():
            for ly in m.decoder.decoder_blocks.layers:
                ly.multihead_attn.out_proj.weight.mul_(8.0)
        return m

    G.build_vitomr = build
    try:
        for seed in range(300, 340):
            G.gen_vitomr(name, dh64, imgs, seed)
            fx = torch.load(os.path.join(OUT, name + ".pt"), map_location="cpu", weights_only=False)
            rows = fx["ref_bf16"]["seqs"]
            diff01 = float((rows[0] != rows[1]).float().mean()), float((rows[0] != rows[2]).float().mean())
            same32 = torch.equal(fx["ref_fp32"]["seqs"], fx["ref_bf16"]["seqs"])
            print(f"seed {seed}: rows differ in {diff01} of their positions; oracle bf16 tokens equal = {fx['oracle_bf16_tokens_equal']}; fp32 == bf16 tokens {same32}")
            if min(diff01) >= 0.5 and fx["oracle_bf16_tokens_equal"]:
                print("kept seed", seed, rows.tolist())
                return
        raise SystemExit("no seed gave distinct rows")
    finally:
        G.build_vitomr = plain_build


if __name__ == "__main__":
    gen_surface_json()
    gen_mae_surface()
    gen_distinct_rows()
