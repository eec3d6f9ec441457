"""the step semantics of the two training loops (`acai_omr_amd/train/loops.py`).  CPU: schedulers, sequence preparation,
checkpoint layouts.  GPU: whole epochs of the HIP path (fused AdamW, LR schedule, accumulation, flush on the last batch) against the same loop
written over the CPU oracle + `torch.optim.AdamW`."""
import math

import pytest
import torch

from conftest import VOCAB, load_golden


def test_tf_scheduler_and_epoch_forms():
    from acai_omr_amd.train.loops import TFConfig, TFScheduler, calc_tau, calc_teacher_forcing_prob
    cfg = TFConfig(1.0, 5.0, False)
    s = TFScheduler(cfg, 1.0, 0.0, 5.0, 0.1, soft_epochs=1, anneal_epochs=2, num_steps_per_epoch=4)   # soft for 4 steps, anneal over 8
    seen = []
    for _ in range(11):
        s.step()
        seen.append((cfg.tf_prob, cfg.tau, cfg.use_hard_sampling))
    # omr_teacher_force_train.py:75-83 by hand: values come from the count BEFORE the increment
    for k, (p, tau, hard) in enumerate(seen):
        prog = k / 8
        assert p == max(1.0 - prog, 0.0) and math.isclose(tau, max(5.0 * (0.1 / 5.0) ** prog, 0.1), rel_tol=1e-12)
        assert hard == (k >= 4)
    assert seen[8][0] == 0.0 and seen[10][1] == 0.1 and s.step_count == 11
    assert calc_teacher_forcing_prob(0, 35, 1.0, 0.0) == 1.0 and calc_teacher_forcing_prob(35, 35, 1.0, 0.0) == 0.0
    assert math.isclose(calc_teacher_forcing_prob(7, 35, 1.0, 0.0), 0.8) and calc_tau(40, 35, 5.0, 0.1) == 0.1
    assert math.isclose(calc_tau(7, 35, 5.0, 0.1), 5.0 * 0.02 ** 0.2)


def test_prepare_lmx_sequence_and_constants():
    from acai_omr_amd.train import loops
    from acai_omr_amd.models.models import OMRDecoder
    dec = OMRDecoder(24, VOCAB, num_layers=1, hidden_dim=16, num_heads=2, mlp_dim=16)
    prep = loops.PrepareLMXSequence(dec.tokens_to_idxs)
    words = [w for w in dec.tokens_to_idxs if not w.startswith("<")][:3]
    ids = prep("  " + "  ".join(words) + " \n")
    assert ids.dtype == torch.int64 and ids.tolist() == [0] + [dec.tokens_to_idxs[w] for w in words] + [2]
    with pytest.raises(KeyError):
        prep("no-such-token")
    # the reference's known answers (tests/test_omr_teacher_force_train.py:22-28), on the full vocabulary
    full = loops.PrepareLMXSequence(OMRDecoder(1, VOCAB, num_layers=1, hidden_dim=16, num_heads=2, mlp_dim=16).tokens_to_idxs)
    assert full("measure key:fifths:-7 time").tolist() == [0, 3, 4, 19, 2]
    assert full("tremolo:4 C1").tolist() == [0, 226, 66, 2]
    # pre_train.py:27-36, omr_teacher_force_train.py:30-57
    assert loops.PRETRAIN == dict(epochs=500, checkpoint_freq=50, base_lr=1.5e-4, min_lr=1e-6, betas=(0.9, 0.95), weight_decay=0.05, warmup_epochs=50,
                                  batch_size=64)
    ft = loops.FINE_TUNE
    assert (ft["epochs"], ft["base_lr"], ft["fine_tune_base_lr"], ft["fine_tune_decay_factor"], ft["weight_decay"], ft["grad_accumulation_steps"]) == \
        (40, 1e-4, 1e-5, 0.9, 0.01, 8) and ft["soft_epochs"] == ft["epochs"] // 2 and ft["tf_anneal_epochs"] == 35
    c = loops.StepCounter()
    c.increment()
    assert c.global_step == 1


def test_checkpoint_layouts_round_trip(tmp_path):
    from acai_omr_amd.train import loops
    from acai_omr_amd.utils import cosine_anneal_with_warmup
    m = torch.nn.Linear(4, 3)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sch = cosine_anneal_with_warmup(opt, 1, 4, 1e-6)
    m(torch.ones(2, 4)).sum().backward()
    opt.step()
    sch.step()
    loops.save_pretraining_state(tmp_path / "a.pth", m, opt, sch)
    loops.save_omr_training_state(tmp_path / "b.pth", m, opt, sch)
    a = torch.load(tmp_path / "a.pth", weights_only=False)
    b = torch.load(tmp_path / "b.pth", weights_only=False)
    assert set(a) == {"mae_state_dict", "optimizer_state_dict", "scheduler_state_dict"}          # pre_train.py:40-44
    assert set(b) == {"vitomr_state_dict", "optimizer_state_dict", "scheduler_state_dict"}       # omr_teacher_force_train.py:98-102
    m2 = torch.nn.Linear(4, 3)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    sch2 = cosine_anneal_with_warmup(opt2, 1, 4, 1e-6)
    loops.load_training_state(tmp_path / "b.pth", m2, opt2, sch2)
    assert torch.equal(m2.weight, m.weight) and opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    assert sch2.state_dict()["last_epoch"] == sch.state_dict()["last_epoch"]


class _Loader(list):
    """What the loops need of a DataLoader: iteration and len()."""


def _max_and_mean_diff(model, sd):
    mx = mean = 0.0
    n = 0
    for name, p in model.named_parameters():
        d = (p.detach().cpu() - sd[name].detach()).abs()
        mx, mean, n = max(mx, float(d.max())), mean + float(d.sum()), n + d.numel()
    return mx, mean / n


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from acai_omr_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_fused_adamw_step_invalidates_operand_copies(dev):
    """The fused step writes parameters from a raw kernel: it must move their version counters, or the cached bf16 / transposed operand copies
    (engine.WeightCache) would keep feeding the GEMMs last step's weights."""
    from acai_omr_amd.engine import WeightCache
    from acai_omr_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.randn(8, 16, device=dev))
    wc = WeightCache()
    w16, wt32 = wc.w(p, "bf16"), wc.wt(p, "fp32")
    assert wc.w(p, "bf16") is w16                                    # cached while untouched
    v0 = p._version
    p.grad = torch.ones_like(p)
    FusedAdamW([p], lr=0.1).step()
    assert p._version > v0
    assert torch.equal(wc.w(p, "bf16"), p.detach().to(torch.bfloat16)) and not torch.equal(wc.w(p, "bf16"), w16)
    assert torch.equal(wc.wt(p, "fp32"), p.detach().t()) and not torch.equal(wc.wt(p, "fp32"), wt32)


@pytest.mark.gpu
def test_pretrain_epochs_vs_oracle_loop(dev):
    """Two epochs x three ragged batches of the small MAE: per-batch optimizer steps, per-epoch LR schedule, returned epoch averages and final
    parameters against the oracle + torch.optim.AdamW under the reference's loop (pre_train.py:46-71); then a validation pass and a checkpoint
    that resumes into stock torch AdamW."""
    import oracle.vitomr_oracle as O
    from acai_omr_amd.models.models import MAE, MAELoss
    from acai_omr_amd.optim import FusedAdamW
    from acai_omr_amd.train import loops
    from acai_omr_amd.utils import cosine_anneal_with_warmup
    fx = load_golden("mae_small")
    cfg = fx["cfg"]
    P = cfg["P"]
    mae = MAE(cfg["mask_ratio"], P, cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"], decoder_hidden_dim=cfg["dec_dim"],
              encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"])
    mae.load_state_dict(fx["state_dict"])
    g = torch.Generator().manual_seed(11)
    shapes = [[(8, 16), (24, 40)], [(12, 20), (16, 16), (4, 8)], [(24, 24)]]
    batches = _Loader([[(im, im) for im in (torch.rand(1, h, w, generator=g) for h, w in b)] for b in shapes])
    noises = [[torch.rand((im.shape[-2] // P) * (im.shape[-1] // P), generator=g) for im, _ in b] for b in batches] * 2
    LR, EPOCHS = 3e-3, 2

    # the reference's loop over the oracle
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in fx["state_dict"].items()}
    names = [n for n, _ in mae.named_parameters()]
    opt_o = torch.optim.AdamW([sd[n] for n in names], lr=LR, betas=(0.9, 0.95), weight_decay=0.05)
    sch_o = cosine_anneal_with_warmup(opt_o, 1, 4, 1e-6)
    avg_o, it = [], iter(noises)
    for _ in range(EPOCHS):
        tot = 0.0
        for b in batches:
            pred, lm, tgt, _ = O.mae_forward(b, next(it), sd, P, cfg["mask_ratio"], cfg["enc_kwargs"]["num_heads"], cfg["dec_kwargs"]["num_heads"], prec="fp32")
            loss = O.mae_loss(pred, lm, tgt)
            tot += loss.item()
            loss.backward()
            opt_o.step()
            opt_o.zero_grad()
        sch_o.step()
        avg_o.append(tot / len(batches))

    class Injected(torch.nn.Module):   # the loop calls mae(batch); the masking noise of each call comes from the same list
        def __init__(self, inner, queue):
            super().__init__()
            self.inner, self.queue = inner, list(queue)

        def forward(self, batch):
            return self.inner(batch, noises=self.queue.pop(0))

    mae = mae.to(dev)
    model = Injected(mae, noises)
    opt = FusedAdamW(mae.parameters(), lr=LR, betas=(0.9, 0.95), weight_decay=0.05)
    sch = cosine_anneal_with_warmup(opt, 1, 4, 1e-6)
    avg = [loops.pretrain_epoch(model, batches, MAELoss(), opt, sch, dev) for _ in range(EPOCHS)]
    # epoch 1 (three optimizer steps in) agrees to fp32 noise; by epoch 2 AdamW's normalised update has amplified ~0 gradients of either side
    assert abs(avg[0] - avg_o[0]) < 1e-5 and abs(avg[1] - avg_o[1]) < 2e-3, (avg, avg_o)
    assert opt.param_groups[0]["lr"] == opt_o.param_groups[0]["lr"]                      # scheduler moved once per epoch
    assert all(float(opt.state[p]["step"]) == EPOCHS * len(batches) for p in mae.parameters() if p in opt.state)
    mx, mean = _max_and_mean_diff(mae, sd)
    assert mx < 2.5 * LR and mean < 0.02 * LR * EPOCHS * len(batches), (mx, mean)       # AdamW: a flipped sign of a ~0 gradient moves 2 lr
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0 for p in mae.parameters())   # zero_grad after every step

    model.queue = list(noises[:3])
    val = loops.pretrain_validation(model, batches, MAELoss(), dev)
    assert not mae.training and math.isfinite(val) and abs(val - avg[-1]) < 0.5

    # checkpoint written by the loop resumes in stock torch AdamW (same state layout) and takes the same next step
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:
        loops.save_pretraining_state(os.path.join(d, "ck.pth"), mae, opt, sch)
        ck = torch.load(os.path.join(d, "ck.pth"), map_location="cpu", weights_only=False)
    assert set(ck) == {"mae_state_dict", "optimizer_state_dict", "scheduler_state_dict"} and set(ck["mae_state_dict"]) == set(fx["state_dict"])
    twin = MAE(cfg["mask_ratio"], P, cfg["pe_h"], cfg["pe_w"], encoder_hidden_dim=cfg["enc_dim"], decoder_hidden_dim=cfg["dec_dim"],
               encoder_kwargs=cfg["enc_kwargs"], decoder_kwargs=cfg["dec_kwargs"]).to(dev)
    twin.load_state_dict(ck["mae_state_dict"])
    topt = torch.optim.AdamW(twin.parameters(), lr=LR, betas=(0.9, 0.95), weight_decay=0.05)
    topt.load_state_dict(ck["optimizer_state_dict"])
    assert topt.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    for net, o in ((mae, opt), (twin, topt)):
        net.train()
        pred, lm, tgt = net(batches[0], noises=noises[0])
        MAELoss()(pred, lm, tgt).backward()
        o.step()
    for (n, a), (_, b) in zip(mae.named_parameters(), twin.named_parameters()):
        assert (a - b).abs().max() < 1e-5, n


@pytest.mark.gpu
def test_fine_tune_epoch_vs_oracle_loop(dev):
    """One epoch x three batches, accumulation 2 (so: a step after batch 2 and the flush after batch 3), bf16 autocast, layer-wise-LR param
    groups, per-step LR + TF schedulers, counter and writer - against the reference's loop (omr_teacher_force_train.py:104-142) over the
    oracle's bf16 restatement + torch AdamW.  tf_prob is pinned to 1 (min = initial), where forward_train is the teacher-forced pass."""
    import oracle.vitomr_oracle as O
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    from acai_omr_amd.optim import FusedAdamW
    from acai_omr_amd.train import loops
    from acai_omr_amd.utils import cosine_anneal_with_warmup
    fx = load_golden("tf_small")
    cfg = fx["cfg"]
    P = cfg["P"]

    def build():
        enc = FineTuneOMREncoder(P, cfg["pe_h"], cfg["pe_w"], cfg["ft_depth"], num_layers=cfg["enc_layers"], hidden_dim=cfg["enc_dim"],
                                 num_heads=cfg["enc_heads"], mlp_dim=cfg["enc_mlp"], transformer_dropout=0.0)
        dec = OMRDecoder(cfg["max_len"], VOCAB, num_layers=cfg["dec_layers"], hidden_dim=cfg["dec_dim"], num_heads=cfg["dec_heads"], mlp_dim=cfg["dec_mlp"],
                         transformer_dropout=0.0)
        m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dim=cfg["head_dim"], transition_head_dropout=0.0)
        m.load_state_dict(fx["state_dict"])
        return m

    m = build()
    g = torch.Generator().manual_seed(21)

    def sample(h, w, n):
        return torch.rand(1, h, w, generator=g), torch.cat([torch.tensor([0]), torch.randint(3, 227, (n,), generator=g), torch.tensor([2])])

    batches = _Loader([[sample(8, 16, 5), sample(24, 40, 9)], [sample(12, 20, 3), sample(16, 16, 7), sample(4, 8, 2)], [sample(24, 24, 11)]])
    BASE, FT, DECAY, ACC = 2e-3, 1e-3, 0.9, 2

    def groups_for(model, params_of):
        gs, _ = model.create_fine_tune_param_groups(BASE, FT, DECAY)
        ids = {id(p): n for n, p in model.named_parameters()}
        return [{"params": [params_of(ids[id(p)]) for p in list(gr["params"])], "lr": gr["lr"]} for gr in gs]

    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in fx["state_dict"].items()}
    opt_o = torch.optim.AdamW(groups_for(m, lambda n: sd[n]), betas=(0.9, 0.95), weight_decay=0.01)
    steps_per_epoch = -(len(batches) // -ACC)
    sch_o = cosine_anneal_with_warmup(opt_o, 1, 3, 1e-6, num_train_batches=steps_per_epoch)
    tot, n_steps = 0.0, 0
    for i, b in enumerate(batches):
        pred, tgt = O.teacher_forced_forward(b, sd, cfg["enc_heads"], cfg["dec_heads"], P, "bf16")
        loss = O.ce_loss(pred, tgt, 1)
        tot += loss.item()
        loss.backward()
        if (i + 1) % ACC == 0 or i + 1 == len(batches):
            opt_o.step()
            opt_o.zero_grad()
            sch_o.step()
            n_steps += 1
    avg_o = tot / len(batches)

    m = m.to(dev)
    named = dict(m.named_parameters())
    opt = FusedAdamW(groups_for(m, lambda n: named[n]), betas=(0.9, 0.95), weight_decay=0.01)
    sch = cosine_anneal_with_warmup(opt, 1, 3, 1e-6, num_train_batches=steps_per_epoch)
    tfc = loops.TFConfig(1.0, 5.0, False)
    tfs = loops.TFScheduler(tfc, 1.0, 1.0, 5.0, 0.1, soft_epochs=1, anneal_epochs=2, num_steps_per_epoch=steps_per_epoch)
    counter = loops.StepCounter()

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, step):
            self.rows.append((tag, float(value), step))

    w = Writer()
    avg = loops.fine_tune_epoch(m, batches, OMRCELoss(m.decoder.pad_idx), opt, sch, "cuda", ACC, tfc, tfs, w, counter)
    assert abs(avg - avg_o) < 5e-3, (avg, avg_o)
    assert n_steps == 2 and counter.global_step == 2 and tfs.step_count == 2 and tfc.tf_prob == 1.0 and tfc.tau < 5.0 and not tfc.use_hard_sampling
    assert [gr["lr"] for gr in opt.param_groups] == [gr["lr"] for gr in opt_o.param_groups]     # per-optimizer-step schedule on every group
    assert len(opt.param_groups) == 2 + cfg["ft_depth"] + 3
    tags = [r[0] for r in w.rows]
    assert tags.count("train/loss") == 2 and [r[2] for r in w.rows if r[0] == "train/loss"] == [0, 1]
    assert [r[1] for r in w.rows if r[0] == "train/hyperparams/fine_tune_base_lr"][-1] == opt.param_groups[2]["lr"]
    mx, mean = _max_and_mean_diff(m, sd)
    assert mx < 2.5 * BASE * n_steps and mean < 0.1 * BASE * n_steps, (mx, mean)
    val = loops.fine_tune_validation(m, batches, OMRCELoss(m.decoder.pad_idx), "cuda")
    assert not m.training and math.isfinite(val)
