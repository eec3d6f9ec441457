import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from conftest import load_golden
from test_gpu_parity import build_vitomr
from acai_omr_amd import engine as EG
fx = load_golden("vitomr_dh64"); cfg = fx["cfg"]
m = build_vitomr(cfg, fx["state_dict"], "cuda", torch.bfloat16, max_batch=32)
with torch.no_grad():
    lat, mask = m.encoder(fx["imgs"]); mem = m.transition_head(lat)
    mem32, lens = EG.unpad_rows(mem, mask)
    blocks = m.decoder.decoder_blocks
    G = 3
    outs = {}
    for mode in ("flat", "group"):
        if mode == "group":
            blocks.prepare_caches_packed(mem32, None, lens, group_size=G)
        else:
            rep = torch.cat([mem32[sum(lens[:i]):sum(lens[:i+1])] for i in range(len(lens)) for _ in range(G)])
            blocks.prepare_caches_packed(rep, None, [l for l in lens for _ in range(G)])
        eng = blocks.engine(mem32.device)
        tok = torch.arange(eng.B, device="cuda") % 7 + 3
        lg1 = eng.logits_step(tok, 1).clone()
        lg2 = eng.logits_step(tok + 1, 2).clone()
        outs[mode] = (lg1, lg2)
    for i in range(2):
        d = (outs["flat"][i] - outs["group"][i]).abs()
        print("step", i + 1, "max diff", float(d.max()), "mean", float(d.mean()), "ref abs mean", float(outs["flat"][i].abs().mean()), "rows", d.max(dim=1).values.tolist())
    print("lens", lens, "B", eng.B, "nsplit", eng.cross_nsplit)
