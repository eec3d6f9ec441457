"""Benchmark of the hot path on MI355X: LMX tokens/sec of KV-cached greedy decode (BASELINE.json metric).

    python bench.py --gpus 1 --steps 256 --warmup 32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): full-size ViTOMR (FineTuneOMREncoder ViT-B/16 + 12-layer d=1024 OMRDecoder, random init,
seed 0), batch of 8 synthetic 512x2048 system images per GPU (4096 patches each), reference inference plumbing
(encoder fp32, transition head + decode in bf16 with a bf16 KV cache).  A "step" is one greedy decode step of the
whole batch (8 tokens per GPU), replayed from one captured hipGraph; inputs (cross K/V, weights) are resident in HBM
before the timed region.  Encoder / head / cross-K/V prefill are timed once and reported separately (`prefill_ms`).
Multi-GPU: images are independent -> each rank decodes its own batch of 8 (weak scaling), no data-path collective.

One JSON line on rank 0 with `roofline` (dominant kernel: the cross-attention K/V stream, timed live with HIP events
on the launch stream) and `cpu_baseline` (the CPU oracle on the host cores, bounded sample)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def build_model(device, batch, cache_dtype=torch.bfloat16):
    from acai_omr_amd.inference.vitomr_inference import set_up_omr_inference
    torch.manual_seed(0)
    vitomr, _ = set_up_omr_inference(os.path.join(ROOT, "lmx_vocab.txt"), max_batch_size=batch, cache_dtype=cache_dtype, device=device)
    return vitomr.eval()


def time_cross_attn_kernel(eng, iters=48):
    """Average duration of ONE launch of the dominant kernel (decode_attn_kernel<bf16, 8, RAGGED>) exactly as a decode step launches it
    (split over the memory, partials merged inside the launch by the last-arriving workgroup), cycling over the 12 layers' K/V so that every
    launch streams from HBM as it does inside a step (12 x 134 MB >> 256 MB Infinity Cache).  HIP events on the launch stream."""
    import ctypes

    from acai_omr_amd import _lib, ops
    L = _lib.lib()
    q = torch.randn(eng.B, 3 * eng.E, device=eng.device)
    out = eng.ws["attn"]
    args = lambda l: (q.data_ptr(), q.stride(0), eng.k_cross[l].data_ptr(), eng.v_cross[l].data_ptr(), eng.cross_off.data_ptr(),  # noqa: E731
                      eng.cross_len.data_ptr(), eng.partial.data_ptr(), out.data_ptr(), out.stride(0), eng.B, eng.H, eng.dh, eng.dhp, eng.CROSS_CHUNK,
                      eng.cross_nsplit, _lib.ACAI_BF16 if eng.bf else _lib.ACAI_F32, 1 if eng.bf else 0, eng.tickets.data_ptr(), ops._st())
    for l in range(eng.L):
        _lib.check(L.acai_decode_attn(*args(l)), "acai_decode_attn")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        _lib.check(L.acai_decode_attn(*args(i % eng.L)), "acai_decode_attn")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3  # seconds per launch


def cpu_baseline(vitomr, lens, steps):
    """The CPU oracle (oracle/vitomr_oracle.py, "port") on the host cores: same weights, same batch shape, decode only."""
    from oracle import vitomr_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # the GPU box gives one GPU's job a 16-CPU share; more threads than that only adds contention
    torch.set_num_threads(cores)
    sd = {"decoder." + k: O.rbf16(v.detach().float().cpu()) if v.dim() >= 1 and "norm" not in k and "embedding" not in k else v.detach().float().cpu()
          for k, v in vitomr.decoder.state_dict().items()}
    O.WEIGHTS_PREROUNDED = True
    try:
        g = torch.Generator().manual_seed(0)
        mem = O.rbf16(torch.randn(sum(lens), vitomr.decoder.hidden_dim, generator=g))
        state = O.DecodeState(mem, lens, sd, vitomr.decoder.num_heads, "bf16")   # cross K/V prefill, untimed (as on the GPU)
        B = len(lens)
        tok = torch.zeros(B, dtype=torch.long)
        logits = O.decode_step(state, tok, 1)  # warm-up step
        t0 = time.perf_counter()
        for t in range(2, 2 + steps):
            tok, _ = O.next_token(logits, "bf16")
            logits = O.decode_step(state, tok, t)
        dt = time.perf_counter() - t0
    finally:
        O.WEIGHTS_PREROUNDED = False
    return dict(value=B * steps / dt, unit="tokens/s", cores=cores, kind="port",
                sample=f"{steps} greedy decode steps x {B} sequences (S={lens[0]}) after an untimed cross-K/V prefill; CPU oracle in its autocast(bf16) restatement, weights rounded once")


def bench_mae(dev, rank, world, dist, batch, height, width, steps, dtype):
    """Second half of the BASELINE.json metric: MAE images/sec = images / (forward + MAELoss + backward [+ DP gradient
    all-reduce]) on `batch` synthetic HxW images per GPU, full-size MAE(0.75, 16, 60, 200) (pre_train.py:156-159).
    The step ends with the optimizer update (pre_train.py:59-61: AdamW lr 1.5e-4, betas (0.9, 0.95), weight decay 0.05) through the fused
    multi-tensor AdamW kernel."""
    from torch.amp import autocast

    from acai_omr_amd.config import MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
    from acai_omr_amd.dist import GradAllReduce, global_mean_scale
    from acai_omr_amd.models.models import MAE, MAELoss
    from acai_omr_amd.optim import FusedAdamW
    torch.manual_seed(0)
    mae = MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH).to(dev).train()
    ddp = GradAllReduce(mae) if world > 1 else None
    opt = FusedAdamW(mae.parameters(), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
    g = torch.Generator().manual_seed(2000 + rank)
    imgs = [torch.rand(1, height, width, generator=g).to(dev) for _ in range(batch)]
    data = list(zip(imgs, imgs))
    loss_fn = MAELoss()

    def step():
        if ddp is not None:
            ddp.zero_grad()
        else:
            mae.zero_grad(set_to_none=True)
        with autocast(device_type="cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
            pred, loss_mask, target, _ = mae.forward_packed(data)
        loss = loss_fn(pred, loss_mask, target)
        if ddp is not None:
            loss = loss * global_mean_scale(float(loss_mask.sum().item()), device=dev)  # mean over the GLOBAL masked count
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        return loss

    step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flop_img = 2.29e12 * (height * width) / (512 * 2048) if (height, width) == (512, 2048) else None
    out = dict(images_per_s=world * batch * steps / dt, ms_per_step=dt / steps * 1e3, batch_per_gpu=batch, image=f"{height}x{width}", dtype=dtype,
               steps=steps, includes="forward + MAELoss + backward" + (" + RCCL gradient all-reduce" if world > 1 else "") + " + fused AdamW step", loss=float(loss.detach()))
    if flop_img:
        out["tflops_algorithmic"] = flop_img * world * batch * steps / dt / 1e12
    del mae, ddp, opt
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--cpu-steps", type=int, default=192, help="timed CPU-oracle decode steps (192 x 8 tokens is ~10 s on 16 cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encoder-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--mae-batch", type=int, default=32)
    ap.add_argument("--mae-steps", type=int, default=3)
    ap.add_argument("--mae-dtype", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--no-mae", action="store_true")
    a = ap.parse_args()

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    local = local % max(1, torch.cuda.device_count())   # (rehearsals may put several ranks on one card)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ACAI_BENCH_BACKEND", "nccl")   # "nccl" is RCCL over xGMI on ROCm; "gloo" only for single-card rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from torch.amp import autocast
    vitomr = build_model(dev, a.batch)
    g = torch.Generator().manual_seed(1000 + rank)
    imgs = [torch.rand(1, a.height, a.width, generator=g).to(dev) for _ in range(a.batch)]

    # ---- prefill (timed once, reported separately) ----------------------------------------------------------------------
    with torch.no_grad():
        def prefill():
            with autocast(device_type="cuda", dtype=torch.bfloat16, enabled=a.encoder_dtype == "bf16"):
                lat32, _, lens = vitomr.encoder.forward_packed(imgs)
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                mem = vitomr.transition_head.forward_packed(lat32)
            vitomr.decoder.decoder_blocks.prepare_caches_packed(None, mem, lens)
            return lens
        prefill()  # warm-up (code-object loads, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lens = prefill()
        torch.cuda.synchronize()
        prefill_s = time.perf_counter() - t0

    eng = vitomr.decoder.decoder_blocks.engine(dev)
    cap = eng.Tmax - 2   # decode steps one armed sequence can take before the 1536-token self-attention cache is full
    pos = 0

    def run_steps(n):
        """Enqueue n decode steps; when the cache is full the device-side state is re-armed (a fresh <bos>) and decoding goes on."""
        nonlocal pos
        while n > 0:
            if pos >= cap:
                eng.arm(eng.B)
                pos = 0
            m = min(n, cap - pos)
            eng.launch_steps(m)
            pos += m
            n -= m

    cur = torch.cuda.current_stream(dev)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        eng.arm(eng.B)
        eng.ensure_graph(1)
        eng.ensure_graph(eng.STEPS_PER_GRAPH)
        eng.arm(eng.B)
        run_steps(a.warmup)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(a.steps)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    cur.wait_stream(eng.stream)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tokens = world * a.batch * a.steps
    assert int(eng.step[0].item()) == 1 + pos  # every replay advanced the device-side position

    mae_res = None
    if not a.no_mae:
        try:
            mae_res = bench_mae(dev, rank, world, dist, a.mae_batch, a.height, a.width, a.mae_steps, a.mae_dtype)
        except Exception as e:   # the secondary leg must not cost the headline line (every rank runs the same code: an error is symmetric)
            mae_res = dict(error=f"{type(e).__name__}: {e}")

    out = None
    if rank == 0:
        S = lens[0]
        t_mid = min(a.warmup + a.steps // 2, cap // 2) if a.warmup + a.steps <= cap else cap // 2
        # algorithmic bytes: weights once per step + per sequence 12*2*(S + t)*1024*2 B
        w_bytes = sum(p.numel() for n, p in vitomr.decoder.named_parameters() if "decoder_blocks.layers" in n and p.dim() == 2) * 2 + \
            vitomr.decoder.unembed.weight.numel() * 2
        step_bytes = w_bytes + a.batch * 12 * 2 * (S + t_mid) * 1024 * 2
        k_s = time_cross_attn_kernel(eng)
        k_bytes = sum(lens) * 2 * eng.E * 2   # K and V rows of every sequence, bf16, one layer
        # HBM traffic of the same kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950
        # correction applied - profiles/r01_pmc_cross_attn.json); only quoted when it was measured on this workload shape
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_cross_attn.json")
        if os.path.exists(pmc) and a.batch == 8 and S == 4096:
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
        roof = dict(bound="hbm", kernel="decode_attn_kernel<bf16,8,RAGGED> (cross-attention K/V stream, one layer, all sequences)",
                    achieved=k_bytes / k_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=k_bytes / k_s / 1e9 / HBM_PEAK_GBS, traffic=traffic,
                    kernel_us=k_s * 1e6, bytes_per_launch=k_bytes,
                    step_bytes=step_bytes, step_achieved_GBs=step_bytes / (dt / a.steps) / 1e9)
        out = dict(metric="LMX tokens/sec (greedy decode, KV cache)", value=tokens / dt, unit="tokens/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                   config=dict(workload=f"vitomr_greedy_decode batch {a.batch}/GPU of {a.height}x{a.width} images ({S} patches), decode steps {a.warmup + 1}..{a.warmup + a.steps}" + ("" if a.warmup + a.steps <= cap else f" (re-armed every {cap} steps)"),
                               batch_per_gpu=a.batch, memory_len=S, decoder="12 x d1024 h16 mlp4096, V=227", hipgraph=True),
                   prefill_ms=prefill_s * 1e3, prefill_encoder_dtype=a.encoder_dtype, roofline=roof, mae=mae_res)
        # decode-only (`value`) and end to end.  Composed from the two measured parts - the prefill timed once above and the timed
        # decode steps - for a generation of warmup + steps tokens per sequence; not a separately timed run.
        gen = a.warmup + a.steps
        out["end_to_end"] = dict(tokens_per_s=world * a.batch * gen / (prefill_s + gen * dt / a.steps), generated_tokens_per_sequence=gen,
                                 includes="encoder (fp32) + transition head + cross-K/V prefill + decode steps" if a.encoder_dtype == "fp32"
                                 else "encoder (bf16) + transition head + cross-K/V prefill + decode steps")
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vitomr, lens, a.cpu_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
