"""Benchmark of the hot path on MI355X (BASELINE.json metric: LMX tokens/sec (greedy decode) + MAE images/sec, 512x2048 input).

    python bench.py --gpus 1 --steps 256 --warmup 32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Headline (`value`): full-size ViTOMR (FineTuneOMREncoder ViT-B/16 + 12-layer d=1024 OMRDecoder, random init, seed 0), batch of 8 synthetic
512x2048 system images per GPU (4096 patches each), reference inference plumbing (encoder fp32, transition head + decode in bf16 with a bf16
KV cache).  A "step" is one greedy decode step of the whole batch (8 tokens per GPU) replayed from a captured hipGraph; inputs (cross K/V,
weights) are resident in HBM before the timed region.  Multi-GPU: images are independent -> each rank decodes its own batch (weak scaling, no
data-path collective).

Further legs in the same JSON line (each a dict; an exception in a leg is reported in it and never costs the headline):
  end_to_end     inference() TIMED AS CALLED on the headline batch (8 x 512x2048): encoder + head + prefill + 288 generated tokens      (N = 1)
  latency_b1     BASELINE config 1 on the GPU: inference() on ONE 256x1024 image, 256 generated tokens, ms/token, with the CPU oracle's
                 run of the same call beside it                                                                                    (N = 1)
  mae            config 2: MAE fwd + MAELoss + bwd + fused AdamW, batch 32 x 512x2048 per GPU, bf16 (+ RCCL gradient all-reduce when N > 1),
                 with its own `roofline` (MFMA bound; whole step and the dominant attention kernel timed live) and `cpu_baseline`
  tf_step        config 3: ScheduledSamplingViTOMR.forward_train + OMRCELoss + bwd, batch 16 x (512x2048, T = 512), bf16          (N = 1)
  ragged_decode  config 4: greedy decode of the ragged batch of 8 systems 256x1024 ... 768x3072, 512 steps, hipGraph              (N = 1)
  config5        config 5: global batch of 32 N ragged images dealt to the ranks by cost, DP MAE step + DP teacher-forced step with the
                 exposed all-reduce time, and `dp_parity_max_abs_diff` (DP step == single-process global-batch step, tiny fixtures)  (N > 1)

`roofline` (dominant kernel of the headline: the cross-attention K/V stream, timed live with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle on the host cores, bounded sample) as the harness contract asks."""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
CONFIG4_SHAPES = [(256, 1024), (256, 2048), (384, 1536), (512, 2048), (512, 3072), (640, 2560), (768, 2304), (768, 3072)]


# ---- algorithmic work (a multiply-add = 2 FLOP; the causal half of decoder self-attention counted in full) -------------------
def flops_vit_stack(n, d, mlp, layers):
    return layers * (8 * n * d * d + 4 * n * n * d + 4 * n * d * mlp)


def flops_encoder_fwd(n):
    return 2 * n * 256 * 768 + flops_vit_stack(n, 768, 3072, 12)


def flops_mae_fwd(n):
    """MAE forward per image of n patches: encoder on the kept quarter + decoder_embed + 8-layer d=512 decoder on all n + unembed."""
    k = n // 4
    return flops_encoder_fwd(k) + 2 * k * 768 * 512 + flops_vit_stack(n, 512, 3072, 8) + 2 * n * 512 * 256


def flops_tf_fwd(n, t):
    """Teacher-forced ViTOMR forward per image: encoder + transition head + TWO decoder passes (forward_train) of t tokens over n patches."""
    head = 2 * n * (768 * 4096 + 4096 * 1024)
    dec = 12 * (8 * t * 1024 ** 2 + 4 * t * t * 1024 + 4 * t * 1024 ** 2 + 4 * n * 1024 ** 2 + 4 * t * n * 1024 + 4 * t * 1024 * 4096) + 2 * t * 1024 * 227
    return flops_encoder_fwd(n) + head + 2 * dec


def host_cores(cap=16):
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(cores, cap)   # the GPU box gives one GPU's job a 16-CPU share; more threads than that only adds contention


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def decode_weight_bytes(decoder):
    """the bf16 weight bytes ONE decode step streams = 12 layers x 14 d^2 + the unembed.  The K/V rows [E:] of the cross
    attention's in_proj_weight are read by the prefill only (kv_caching.py:212-215 uses rows [:d] in a step) and are not counted:
    352.8 MB for the full-size decoder (counting them gave 403 MB in rounds 1-2)."""
    E = decoder.hidden_dim
    n = 0
    for name, p in decoder.named_parameters():
        if "decoder_blocks.layers" in name and p.dim() == 2:
            n += E * p.shape[1] if name.endswith("multihead_attn.in_proj_weight") else p.numel()
    return (n + decoder.unembed.weight.numel()) * 2


def build_model(device, batch, cache_dtype=torch.bfloat16):
    from acai_omr_amd.inference.vitomr_inference import set_up_omr_inference
    torch.manual_seed(0)
    vitomr, _ = set_up_omr_inference(os.path.join(ROOT, "lmx_vocab.txt"), max_batch_size=batch, cache_dtype=cache_dtype, device=device)
    return vitomr.eval()


def time_launches(fn, n_warm, iters):
    """Average seconds per call of fn(i), HIP events on the current (launch) stream."""
    for i in range(n_warm):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def time_decode_attn_kernels(eng, iters=48):
    """Average duration of ONE launch of the dominant kernel (decode_attn_kernel<bf16, 8, RAGGED>) exactly as a decode step launches it
    (split over the memory, partials merged inside the launch by the last-arriving workgroup), cycling over the 12 layers' K/V so that every
    launch streams from HBM as it does inside a step (12 x 134 MB >> 256 MB Infinity Cache).  Returns seconds per launch."""
    from acai_omr_amd import _lib, ops
    L = _lib.lib()
    q = torch.randn(eng.B, 3 * eng.E, device=eng.device)
    out = eng.ws["attn"]
    args = lambda l: (q.data_ptr(), q.stride(0), eng.k_cross[l].data_ptr(), eng.v_cross[l].data_ptr(), eng.cross_off.data_ptr(),  # noqa: E731
                      eng.cross_len.data_ptr(), eng.partial.data_ptr(), out.data_ptr(), out.stride(0), eng.B, eng.H, eng.dh, eng.dhp, getattr(eng, "cross_chunk", eng.CROSS_CHUNK),
                      eng.cross_nsplit, _lib.ACAI_BF16 if eng.bf else _lib.ACAI_F32, 1 if eng.bf else 0, eng.tickets.data_ptr(), ops._st())
    return time_launches(lambda i: _lib.check(L.acai_decode_attn(*args(i % eng.L)), "acai_decode_attn"), eng.L, iters)


def cpu_baseline_decode(vitomr, lens, steps):
    """The CPU oracle (oracle/vitomr_oracle.py, "port") on the host cores: same weights, same batch shape, decode only."""
    from oracle import vitomr_oracle as O
    cores = host_cores()
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
    return dict(value=B * steps / dt, unit="tokens/s", cores=cores, cpu=cpu_model(), kind="port",
                sample=f"{steps} greedy decode steps x {B} sequences (S={lens[0]}) after an untimed cross-K/V prefill; CPU oracle in its autocast(bf16) restatement, weights rounded once")


def cpu_baseline_mae(height, width):
    """CPU oracle ("port"), autocast(bf16) restatement: MAE forward + MAELoss + backward (autograd) on ONE image of the benchmarked size."""
    from oracle import vitomr_oracle as O
    from acai_omr_amd.config import MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
    from acai_omr_amd.models.models import MAE
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    mae = MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in mae.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    img = torch.rand(1, height, width, generator=g)
    noise = [torch.rand((height // PATCH_SIZE) * (width // PATCH_SIZE), generator=g)]
    t0 = time.perf_counter()
    pred, lm, tgt, _ = O.mae_forward([(img, img)], noise, sd, PATCH_SIZE, MASK_RATIO, 12, 16, prec="bf16")
    O.mae_loss(pred, lm, tgt).backward()
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="images/s", cores=cores, cpu=cpu_model(), kind="port",
                sample=f"1 image {height}x{width}: forward + MAELoss + backward through the CPU oracle (autocast(bf16) restatement, torch autograd), no optimizer step")


def dist_info(dist, dev, backend):
    """What the line was measured on: the process-group backend as torch reports it ("nccl" = RCCL over xGMI on ROCm, "gloo" = a rehearsal
    without collectives on the GPUs), the world size it saw and every rank's device name - so that a scaling record cannot be mistaken for
    something it is not."""
    name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"
    if dist is None:
        return dict(backend=None, world_size=1, devices=[f"rank 0: {name} ({dev})"])
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, f"rank {dist.get_rank()}: {name} ({dev})")
    return dict(backend=dist.get_backend(), world_size=dist.get_world_size(), devices=names,
                transport="RCCL over xGMI" if dist.get_backend() == "nccl" else f"{dist.get_backend()} (host memory: a rehearsal, not a scaling measurement)")


def _allreduce_label(dist):
    if dist is None:
        return ""
    return " + RCCL gradient all-reduce" if dist.get_backend() == "nccl" else f" + {dist.get_backend()} gradient all-reduce (rehearsal backend, not RCCL)"


def _barrier_sync(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(dt, dist, dev):
    if dist is None:
        return dt
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _timed_steps(step, steps, dist, dev, warm=2):   # (two: the second step is the first that refreshes the cached operand copies in one launch and reuses the allocator's pool)
    for _ in range(warm):
        step()
    _barrier_sync(dist)
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    _barrier_sync(dist)
    return _max_over_ranks(time.perf_counter() - t0, dist, dev), out


# ---- config 2: MAE --------------------------------------------------------------------------------------------------------------------------
def bench_mae(dev, rank, world, dist, batch, height, width, steps, dtype, want_cpu, comm_dtype="fp32"):
    """Second half of the BASELINE.json metric: MAE images/sec = images / (forward + MAELoss + backward [+ DP gradient all-reduce] + AdamW
    (pre_train.py:54-62: lr 1.5e-4, betas (0.9, 0.95), weight decay 0.05, here the fused multi-tensor kernel)) on `batch` synthetic HxW
    images per GPU, full-size MAE(0.75, 16, 60, 200) (pre_train.py:156-159)."""
    from torch.amp import autocast

    from acai_omr_amd import engine as EG
    from acai_omr_amd import ops
    from acai_omr_amd.config import MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
    from acai_omr_amd.dist import GradAllReduce, global_mean_scale
    from acai_omr_amd.models.models import MAE, MAELoss
    from acai_omr_amd.optim import FusedAdamW
    torch.manual_seed(0)
    mae = MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH).to(dev).train()
    ddp = GradAllReduce(mae, comm_dtype=torch.bfloat16 if comm_dtype == "bf16" else torch.float32) if world > 1 else None
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

    dt, loss = _timed_steps(step, steps, dist, dev)
    n = (height // PATCH_SIZE) * (width // PATCH_SIZE)
    step_flops = 3 * flops_mae_fwd(n) * batch          # forward + backward ~ 3 x forward (2.29 TFLOP per 512x2048 image)
    ms = dt / steps * 1e3
    out = dict(images_per_s=world * batch * steps / dt, ms_per_step=ms, batch_per_gpu=batch, image=f"{height}x{width}", dtype=dtype, steps=steps,
               includes="forward + MAELoss + backward" + _allreduce_label(dist if world > 1 else None) + " + fused AdamW step",
               grad_comm_dtype=comm_dtype if world > 1 else None,
               loss=float(loss.detach()), tflops_algorithmic=step_flops / (ms * 1e-3) / 1e12)
    del opt, ddp
    # dominant kernels by time: the d_h = 32 self-attention of the 8-layer MAE decoder over all n patches, timed live exactly as the step issues
    # them (q prescaled by the in-projection's epilogue).  `roofline` is the forward kernel; the backward pair (dQ kernel + dK/dV kernel, the
    # largest single item of the step) is reported beside it against its algorithmic 5 products.
    H, dh = 16, 32
    pre = dtype == "bf16"
    qkv = torch.randn(batch * n, 3 * H * dh, device=dev).to(torch.bfloat16 if dtype == "bf16" else torch.float32)
    cu = EG.cu_from_lens([n] * batch, dev)
    E = H * dh
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    lse = torch.empty(H * batch * n, dtype=torch.float32, device=dev)
    o = ops.attn_varlen(q, k, v, cu, cu, H, dh, n, lse=lse, q_prescaled=pre)
    k_s = time_launches(lambda i: ops.attn_varlen(q, k, v, cu, cu, H, dh, n, lse=lse, q_prescaled=pre), 2, 8)
    dout, dqkv = torch.randn_like(o), torch.empty_like(qkv)
    b_s = time_launches(lambda i: ops.attn_varlen_bwd(q, k, v, o, dout, lse, cu, cu, H, dh, n, n, False, dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:],
                                                      q_prescaled=pre), 2, 6)
    k_flops = 4.0 * batch * H * n * n * dh
    out["roofline"] = dict(bound="mfma", kernel=f"attn_fwd_kernel<{dtype}, d_h=32, q prescaled> (MAE decoder self-attention, one layer, {batch} x {n} tokens)",
                           achieved=k_flops / k_s / 1e12, peak=MFMA_PEAK_TFS, unit="TFLOP/s", frac=k_flops / k_s / 1e12 / MFMA_PEAK_TFS,
                           traffic=None, kernel_us=k_s * 1e6, flops_per_launch=k_flops, step_flops=step_flops,
                           step_achieved_TFs=step_flops / (ms * 1e-3) / 1e12, step_frac=step_flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFS,
                           backward_pair=dict(kernels="attn_bwd_dq2_kernel + attn_bwd_dkv2_kernel (same layer; two lane-owned blocks per wave)", us=b_s * 1e6,
                                              flops_algorithmic=2.5 * k_flops, achieved=2.5 * k_flops / b_s / 1e12,
                                              frac=2.5 * k_flops / b_s / 1e12 / MFMA_PEAK_TFS),
                           note="VALU-issue bound, not MFMA bound: per score the forward issues one exp2 and half a pack, the row sums ride the matrix cores, two query blocks per wave share every fragment read (DESIGN.md section 5)")
    del o, dout, dqkv, lse
    del mae, qkv
    torch.cuda.empty_cache()
    if want_cpu:
        out["cpu_baseline"] = cpu_baseline_mae(height, width)
    return out


# ---- config 3: teacher-forced train step ----------------------------------------------------------------------------------------------------
def bench_tf_step(dev, batch, height, width, T, steps):
    from torch.amp import autocast

    from acai_omr_amd.config import ENCODER_FINE_TUNE_DEPTH, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
    from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    torch.manual_seed(0)
    enc = FineTuneOMREncoder(PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH, ENCODER_FINE_TUNE_DEPTH, transformer_dropout=0.0)
    dec = OMRDecoder(MAX_LMX_SEQ_LEN, os.path.join(ROOT, "lmx_vocab.txt"), num_layers=NUM_DECODER_LAYERS, transformer_dropout=0.0)
    m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dropout=0.0).to(dev).train()
    g = torch.Generator().manual_seed(1)
    data = [(torch.rand(1, height, width, generator=g).to(dev),
             torch.cat([torch.tensor([0]), torch.randint(3, 227, (T,), generator=g), torch.tensor([2])]).to(dev)) for _ in range(batch)]
    loss_fn = OMRCELoss(dec.pad_idx)

    def step():
        m.zero_grad(set_to_none=True)
        with autocast(device_type="cuda", dtype=torch.bfloat16):
            pred, tgt = m.forward_train(data, 0.7, 0.5, False)
        loss = loss_fn(pred, tgt)
        loss.backward()
        return loss

    # two warm-up steps: scheduled sampling draws a different number of positions each step, so the allocator's pool is only settled after a second one
    # (with one, a timed step now and then paid a device allocation: 201 ms against 172)
    dt, loss = _timed_steps(step, steps, None, dev)
    n = (height // PATCH_SIZE) * (width // PATCH_SIZE)
    step_flops = 3 * flops_tf_fwd(n, T + 1) * batch
    ms = dt / steps * 1e3
    out = dict(config="config 3: ScheduledSamplingViTOMR.forward_train(tf_prob 0.7, tau 0.5, soft) + OMRCELoss + backward, bf16 autocast, dropout 0",
               ms_per_step=ms, images_per_s=batch * steps / dt, batch_per_gpu=batch, image=f"{height}x{width}", lmx_tokens=T + 1, steps=steps,
               loss=float(loss.detach()), step_flops=step_flops, tflops_algorithmic=step_flops / (ms * 1e-3) / 1e12,
               mfma_frac=step_flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFS)
    del m, enc, dec, data
    torch.cuda.empty_cache()
    return out


# ---- config 4: ragged decode ----------------------------------------------------------------------------------------------------------------
def bench_ragged_decode(dev, steps, warm=16):
    from torch.amp import autocast
    vitomr = build_model(dev, 8)
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(1, h, w, generator=g).to(dev) for h, w in CONFIG4_SHAPES]
    with torch.no_grad():
        def prefill():
            lat32, _, lens = vitomr.encoder.forward_packed(imgs)
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                mem = vitomr.transition_head.forward_packed(lat32)
            vitomr.decoder.decoder_blocks.prepare_caches_packed(None, mem, lens)
            return lens
        prefill()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lens = prefill()
        torch.cuda.synchronize()
        pf = time.perf_counter() - t0
    eng = vitomr.decoder.decoder_blocks.engine(dev)
    # ACAI_BENCH_NO_GRAPH=1: the same launches enqueued one by one (for `rocprofv3 --pmc` passes only: counter collection over the 8-step
    # graph's ~900 kernel nodes segfaults inside the profiler on this ROCm; the number printed is then not the headline)
    use_graph = os.environ.get("ACAI_BENCH_NO_GRAPH") != "1"
    cur = torch.cuda.current_stream(dev)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        eng.arm(eng.B)
        if use_graph:
            eng.ensure_graph(1)
            eng.ensure_graph(eng.STEPS_PER_GRAPH)
        eng.arm(eng.B)
        eng.launch_steps(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.launch_steps(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    cur.wait_stream(eng.stream)
    w_bytes = decode_weight_bytes(vitomr.decoder)
    step_bytes = w_bytes + sum(12 * 2 * (s + warm + steps // 2) * 1024 * 2 for s in lens)
    out = dict(config=f"config 4: greedy decode, ragged batch of 8 systems 256x1024..768x3072 (sum N = {sum(lens)} patches), {steps} steps, hipGraph",
               tokens_per_s=8 * steps / dt, ms_per_step=dt / steps * 1e3, prefill_ms=pf * 1e3, memory_lens=lens, step_bytes=step_bytes,
               step_achieved_GBs=step_bytes / (dt / steps) / 1e9, hbm_frac=step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS)
    del vitomr, eng
    torch.cuda.empty_cache()
    return out


# ---- config 5: data-parallel MAE + teacher-forced steps on ragged shards ---------------------------------------------------------------------
def bench_config5(dev, rank, world, dist, per_gpu, T, steps, comm_dtype="fp32"):
    """Global batch of per_gpu * world ragged images (the 8 config-4 shapes, repeated) dealt to the ranks by patch count (`shard_by_cost`),
    one DP MAE step (fwd + loss + bwd + gradient all-reduce + AdamW) and one DP teacher-forced step (forward_train + CE + bwd + all-reduce +
    AdamW over the LLRD groups).  `allreduce_exposed_ms` = time the compute stream spends in `GradAllReduce.finish()` per step (HIP events):
    the part of the bucketed all-reduce that backward did not cover."""
    from torch.amp import autocast

    from acai_omr_amd.config import (ENCODER_FINE_TUNE_DEPTH, MASK_RATIO, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT,
                                     PE_MAX_WIDTH)
    from acai_omr_amd.dist import GradAllReduce, dp_parity_check, global_mean_scale, shard_by_cost
    from acai_omr_amd.models.models import MAE, FineTuneOMREncoder, MAELoss, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    from acai_omr_amd.optim import FusedAdamW
    shapes = [CONFIG4_SHAPES[i % 8] for i in range(per_gpu * world)]
    costs = [h * w // 256 for h, w in shapes]
    mine = shard_by_cost(costs, world)[rank]
    g = torch.Generator().manual_seed(3000 + rank)
    imgs = [torch.rand(1, *shapes[i], generator=g).to(dev) for i in mine]
    lmx = [torch.cat([torch.tensor([0]), torch.randint(3, 227, (T,), generator=g), torch.tensor([2])]).to(dev) for _ in mine]
    out = dict(config=f"config 5: global batch {per_gpu * world} ragged images (256x1024..768x3072) dealt by cost, {len(mine)} on rank 0 "
                      f"({sum(costs[i] for i in mine)} patches); DP MAE step + DP teacher-forced step (T = {T + 1})", images_global=per_gpu * world)

    def exposed(ddp, evs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ddp.finish()
        e1.record()
        evs.append((e0, e1))

    # MAE
    torch.manual_seed(0)
    mae = MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH).to(dev).train()
    cdt = torch.bfloat16 if comm_dtype == "bf16" else torch.float32
    wire = 2 if comm_dtype == "bf16" else 4
    ddp = GradAllReduce(mae, comm_dtype=cdt)
    opt = FusedAdamW(mae.parameters(), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
    data = list(zip(imgs, imgs))
    evs = []

    def mae_step():
        ddp.zero_grad()
        with autocast(device_type="cuda", dtype=torch.bfloat16):
            pred, loss_mask, target, _ = mae.forward_packed(data)
        loss = MAELoss()(pred, loss_mask, target) * global_mean_scale(float(loss_mask.sum().item()), device=dev)
        loss.backward()
        exposed(ddp, evs)
        opt.step()
        return loss

    dt, _ = _timed_steps(mae_step, steps, dist, dev)
    torch.cuda.synchronize()
    out["mae"] = dict(ms_per_step=dt / steps * 1e3, images_per_s=per_gpu * world * steps / dt,
                      allreduce_exposed_ms=sum(a.elapsed_time(b) for a, b in evs[-steps:]) / steps,
                      grad_bytes=sum(p.numel() for p in mae.parameters() if p.requires_grad) * 4,
                      grad_bytes_on_wire=sum(p.numel() for p in mae.parameters() if p.requires_grad) * wire, grad_comm_dtype=comm_dtype)
    del mae, ddp, opt, data
    torch.cuda.empty_cache()
    # teacher-forced
    torch.manual_seed(0)
    enc = FineTuneOMREncoder(PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH, ENCODER_FINE_TUNE_DEPTH, transformer_dropout=0.0)
    dec = OMRDecoder(MAX_LMX_SEQ_LEN, os.path.join(ROOT, "lmx_vocab.txt"), num_layers=NUM_DECODER_LAYERS, transformer_dropout=0.0)
    m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dropout=0.0).to(dev).train()
    ddp = GradAllReduce(m, comm_dtype=cdt)
    groups, _ = m.create_fine_tune_param_groups(1e-4, 1e-5, 0.9)
    opt = FusedAdamW(groups, betas=(0.9, 0.95), weight_decay=0.01)
    loss_fn = OMRCELoss(dec.pad_idx)
    data = list(zip(imgs, lmx))
    evs = []

    def tf_step():
        ddp.zero_grad()
        with autocast(device_type="cuda", dtype=torch.bfloat16):
            pred, tgt = m.forward_train(data, 0.7, 0.5, False)
        loss = loss_fn(pred, tgt) * global_mean_scale(float((tgt != dec.pad_idx).sum().item()), device=dev)
        loss.backward()
        exposed(ddp, evs)
        opt.step()
        return loss

    dt, _ = _timed_steps(tf_step, steps, dist, dev)
    torch.cuda.synchronize()
    out["tf_step"] = dict(ms_per_step=dt / steps * 1e3, images_per_s=per_gpu * world * steps / dt,
                          allreduce_exposed_ms=sum(a.elapsed_time(b) for a, b in evs[-steps:]) / steps,
                          grad_bytes=sum(p.numel() for p in m.parameters() if p.requires_grad) * 4,
                          grad_bytes_on_wire=sum(p.numel() for p in m.parameters() if p.requires_grad) * wire, grad_comm_dtype=comm_dtype)
    del m, ddp, opt, data, enc, dec
    torch.cuda.empty_cache()
    out["dp_parity_max_abs_diff"] = dp_parity_check(os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "lmx_vocab.txt"), dev)
    out["dp_parity_ok"] = bool(out["dp_parity_max_abs_diff"] < 1e-4)   # DP step (fp32 buckets) over the process group == the single-process global-batch step
    if not out["dp_parity_ok"]:
        out["error"] = f"data-parallel step differs from the single-process global-batch step by {out['dp_parity_max_abs_diff']:.3e} (>= 1e-4)"
    return out


# ---- timed calls of the entry point itself: inference() as the reference's callers use it ---------------------------------------------------
def _suppress_eos(vitomr):
    """Random-init weights may or may not emit <eos>; a fixed generation length needs it out of reach.  The unembed bias of <eos> is pushed far
    below every other logit (same kernels, same bytes, same step count for every sequence); returns a function that restores it."""
    b = vitomr.decoder.unembed.bias
    old = b.detach()[vitomr.decoder.eos_idx].clone()
    with torch.no_grad():
        b[vitomr.decoder.eos_idx] = -1e4

    def restore():
        with torch.no_grad():
            b[vitomr.decoder.eos_idx] = old
    return restore


def bench_inference_call(dev, shapes, gen, reps, want_cpu, cpu_gen=64):
    """`inference(vitomr, imgs, device, max_inference_len=gen + 1)` (vitomr_inference.py:73-86) TIMED AS CALLED: encoder (fp32) + transition
    head + cross-K/V prefill + the greedy loop with its host poll every 16 tokens + mask_and_clip, on `shapes` synthetic images, <eos>
    suppressed so every sequence generates exactly `gen` tokens.  tokens/s counts generated tokens (excluding <bos>) over the whole call."""
    from acai_omr_amd.inference.vitomr_inference import inference
    vitomr = build_model(dev, len(shapes))
    restore = _suppress_eos(vitomr)
    g = torch.Generator().manual_seed(5)
    imgs = [torch.rand(1, h, w, generator=g).to(dev) for h, w in shapes]
    arg = imgs[0] if len(imgs) == 1 else imgs          # config 1 passes the bare (1,H,W) tensor, as vitomr_inference.py:81 does
    seqs, _, mask = inference(vitomr, arg, "cuda", max_inference_len=gen + 1)     # warm-up: code objects, graph capture, allocator
    torch.cuda.synchronize()
    assert seqs.shape == (len(shapes), gen + 1) and bool(mask.all()), (tuple(seqs.shape), int(mask.sum()))
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seqs, _, mask = inference(vitomr, arg, "cuda", max_inference_len=gen + 1)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    B = len(shapes)
    out = dict(call=f"inference(vitomr, {'img' if B == 1 else f'[{B} imgs]'}, 'cuda', max_inference_len={gen + 1})", images=f"{B} x {shapes[0][0]}x{shapes[0][1]}",
               generated_tokens_per_sequence=gen, seconds_per_call=dt, tokens_per_s=B * gen / dt, ms_per_token=dt / gen * 1e3, reps=reps,
               includes="encoder fp32 + transition head + cross-K/V prefill + greedy loop (hipGraph replays, host poll every 16 tokens) + mask_and_clip; timed around the call, median of reps")
    if want_cpu:
        from oracle import vitomr_oracle as O
        cores = host_cores()
        torch.set_num_threads(cores)
        sd = {k: v.detach().float().cpu() for k, v in vitomr.state_dict().items()}
        cimgs = [im.cpu() for im in imgs]
        t0 = time.perf_counter()
        oseqs, _, _ = O.vitomr_inference(cimgs, sd, 12, 16, 16, cpu_gen + 1)
        cdt = time.perf_counter() - t0
        n_tok = int(oseqs.shape[1] - 1)
        same = bool(torch.equal(oseqs[:, :n_tok + 1], seqs.cpu()[:, :n_tok + 1]))
        out["cpu_baseline"] = dict(value=B * n_tok / cdt, unit="tokens/s (end to end)", seconds=cdt, cores=cores, cpu=cpu_model(), kind="port",
                                   sample=f"the same call through the CPU oracle (encoder fp32, head + decode in its autocast(bf16) restatement), {n_tok} generated tokens",
                                   token_ids_equal_gpu=same)
    restore()
    del vitomr
    torch.cuda.empty_cache()
    return out


def _leg(fn, *a, **kw):
    """A secondary leg must not cost the headline line (every rank runs the same code: an error is symmetric)."""
    try:
        return fn(*a, **kw)
    except Exception as e:
        return dict(error=f"{type(e).__name__}: {e}")


def dry_run(a):
    """`--dry-run`: everything bench.py does AROUND the kernels, executed for real on the CPU with the gloo backend - the rendezvous from the
    launcher's environment, --gpus against WORLD_SIZE, the leg selection per world size, barrier + max-over-ranks around each timed region, the
    ragged shard deal of config 5 (`dist.shard_by_cost`), the global-count loss scaling (`dist.global_mean_scale`) and the one JSON line rank 0
    prints - so that the first multi-GPU run of the driver does not meet an untested code path there.  Every measured quantity is a stand-in."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    dev = torch.device("cpu")
    legs = set((a.legs.split(",") if a.legs is not None else (["e2e", "b1", "mae", "tf", "ragged"] if world == 1 else ["mae", "config5"])))
    legs.discard("")
    if a.legs is None:
        assert legs == ({"e2e", "b1", "mae", "tf", "ragged"} if world == 1 else {"mae", "config5"}), legs   # what the driver's N = 1, 2, 4, 8 runs select
    if a.no_mae:
        legs.discard("mae")
    dinfo = dist_info(dist, dev, None)

    def sync():
        if dist is not None:
            dist.barrier()

    def timed(fn, n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt = timed(lambda: time.sleep(1e-4), a.steps)
    S = (a.height // 16) * (a.width // 16)
    out = dict(metric="LMX tokens/sec (greedy decode, KV cache)", value=world * a.batch * a.steps / dt, unit="tokens/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic", dry_run=True,
               config=dict(workload=f"DRY RUN (no kernels): vitomr_greedy_decode batch {a.batch}/GPU of {a.height}x{a.width} images ({S} patches)", batch_per_gpu=a.batch,
                           memory_len=S, decoder="12 x d1024 h16 mlp4096, V=227", hipgraph=True),
               roofline=None, cpu_baseline=None, mae=None, tf_step=None, ragged_decode=None, config5=None, end_to_end=None, latency_b1=None, dist=dinfo,
               value_steps256=None)
    if "mae" in legs:
        d = timed(lambda: time.sleep(1e-4), a.mae_steps)
        out["mae"] = dict(images_per_s=world * a.mae_batch * a.mae_steps / d, ms_per_step=d / a.mae_steps * 1e3, batch_per_gpu=a.mae_batch, dtype=a.mae_dtype, dry_run=True,
                          includes="forward + MAELoss + backward" + _allreduce_label(dist) + " + fused AdamW step", grad_comm_dtype=a.grad_comm_dtype if world > 1 else None)
    for key, name in (("tf", "tf_step"), ("ragged", "ragged_decode"), ("e2e", "end_to_end"), ("b1", "latency_b1")):
        if key in legs and world == 1:
            out[name] = dict(dry_run=True)
    if "config5" in legs:
        from acai_omr_amd.dist import global_mean_scale, shard_by_cost
        shapes = [CONFIG4_SHAPES[i % 8] for i in range(32 * world)]
        costs = [h * w // 256 for h, w in shapes]
        shards = shard_by_cost(costs, world)
        mine = shards[rank]
        scale = float(global_mean_scale(float(sum(costs[i] for i in mine)), device=dev)) if dist is not None else 1.0
        tot = torch.tensor([scale], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tot)      # the local / global count fractions of all ranks sum to one
        out["config5"] = dict(config=f"DRY RUN: global batch {32 * world} ragged images dealt by cost, {len(mine)} on rank 0 ({sum(costs[i] for i in mine)} patches)",
                              images_global=32 * world, shard_sizes=[len(s) for s in shards], shard_patches=[sum(costs[i] for i in s) for s in shards],
                              count_fractions_sum=float(tot.item()),
                              mae=dict(dry_run=True, ms_per_step=None, images_per_s=None, allreduce_exposed_ms=None, grad_bytes=None, grad_bytes_on_wire=None, grad_comm_dtype=a.grad_comm_dtype),
                              tf_step=dict(dry_run=True, ms_per_step=None, images_per_s=None, allreduce_exposed_ms=None, grad_bytes=None, grad_bytes_on_wire=None, grad_comm_dtype=a.grad_comm_dtype),
                              dp_parity_max_abs_diff=None, dp_parity_ok=None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _split_merge_path():
    """Which way the decode step merges its attention splits on this box (acai_decode_merge_in_launch): a silent change of path is a 3 % change
    of the headline."""
    try:
        from acai_omr_amd import _lib
        return "in-launch" if _lib.lib().acai_decode_merge_in_launch(_lib.ACAI_BF16, 64) == 1 else "separate combine launch"
    except Exception as e:   # (never costs the line)
        return f"unknown ({type(e).__name__})"


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
    ap.add_argument("--mae-steps", type=int, default=5)
    ap.add_argument("--mae-dtype", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--legs", default=None, help="comma list of e2e,b1,mae,tf,ragged,config5 (default: e2e,b1,mae,tf,ragged at N = 1; mae,config5 at N > 1)")
    ap.add_argument("--no-mae", action="store_true")
    ap.add_argument("--grad-comm-dtype", default="fp32", choices=["fp32", "bf16"], help="what the gradient buckets travel as at N > 1 (dist.GradAllReduce comm_dtype)")
    ap.add_argument("--leg-timeout", type=float, default=420.0, help="seconds the secondary legs may take before the line is printed without them")
    ap.add_argument("--dry-run", action="store_true", help="control flow only, on the CPU over gloo: argument parsing, rank / leg selection, the collectives "
                    "around the timed regions and the JSON line's shape, with stand-in numbers (no kernel runs; `dry_run: true` marks the line)")
    a = ap.parse_args()
    if a.dry_run:
        return dry_run(a)

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
    legs = set((a.legs.split(",") if a.legs is not None else (["e2e", "b1", "mae", "tf", "ragged"] if world == 1 else ["mae", "config5"])))
    legs.discard("")
    if a.no_mae:
        legs.discard("mae")
    dinfo = dist_info(dist, dev, None)

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
            eng.launch_steps(m, use_graph=use_graph)
            pos += m
            n -= m

    # ACAI_BENCH_NO_GRAPH=1: the same launches enqueued one by one (for `rocprofv3 --pmc` passes only: counter collection over the 8-step
    # graph's ~900 kernel nodes segfaults inside the profiler on this ROCm; the number printed is then not the headline)
    use_graph = os.environ.get("ACAI_BENCH_NO_GRAPH") != "1"
    cur = torch.cuda.current_stream(dev)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        eng.arm(eng.B)
        if use_graph:
            eng.ensure_graph(1)
            eng.ensure_graph(eng.STEPS_PER_GRAPH)
        eng.arm(eng.B)
        run_steps(a.warmup)
        _barrier_sync(dist)
        t0 = time.perf_counter()
        run_steps(a.steps)
        _barrier_sync(dist)
        dt = time.perf_counter() - t0
        # the same region over 256 steps (0.2 s): the driver's --steps 20 times 14 ms, this side field says what a longer window gives
        dt256 = None
        if a.steps != 256:
            _barrier_sync(dist)
            t1 = time.perf_counter()
            run_steps(256)
            _barrier_sync(dist)
            dt256 = time.perf_counter() - t1
    cur.wait_stream(eng.stream)
    dt = _max_over_ranks(dt, dist, dev)
    if dt256 is not None:
        dt256 = _max_over_ranks(dt256, dist, dev)
    tokens = world * a.batch * a.steps
    assert int(eng.step[0].item()) == 1 + pos  # every replay advanced the device-side position

    roof = None
    S = lens[0]
    if rank == 0:
        t_mid = min(a.warmup + a.steps // 2, cap // 2) if a.warmup + a.steps <= cap else cap // 2
        # algorithmic bytes: weights once per step + per sequence 12*2*(S + t)*1024*2 B
        w_bytes = decode_weight_bytes(vitomr.decoder)
        step_bytes = w_bytes + a.batch * 12 * 2 * (S + t_mid) * 1024 * 2
        step_s = dt / a.steps
        k_s = time_decode_attn_kernels(eng)
        k_bytes = sum(lens) * 2 * eng.E * 2   # K and V rows of every sequence, bf16, one layer
        # HBM traffic of the same kernel: NOT measured in this run - the value of the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
        # separate runs, gfx950 correction applied); only quoted when they were taken on this workload shape
        traffic, traffic_src = None, None
        for name in ("r04_pmc_cross_attn.json", "r02_pmc_cross_attn.json", "r01_pmc_cross_attn.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc) and a.batch == 8 and S == 4096:
                traffic, traffic_src = json.load(open(pmc))["hbm_bytes_per_launch"], "profiles/" + name + " (static: rocprofv3 --pmc passes of this workload, not collected in this run)"
                break
        # everything of a step that is not an attention launch is the GEMV chain (weights once per step): derived, from the live-timed launches
        chain_s = max(step_s - eng.L * k_s, 1e-9)
        roof = dict(bound="hbm", kernel="decode_attn_kernel<bf16,8,RAGGED> (cross-attention K/V stream, one layer, all sequences)",
                    achieved=k_bytes / k_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=k_bytes / k_s / 1e9 / HBM_PEAK_GBS, traffic=traffic,
                    traffic_static_from=traffic_src, kernel_us=k_s * 1e6, bytes_per_launch=k_bytes,
                    step_bytes=step_bytes, step_achieved_GBs=step_bytes / step_s / 1e9, step_frac=step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                    rest_of_step=dict(what="step time minus the 12 live-timed cross-attention launches: GEMV chain (decoder weights, once per step) + "
                                           "self-attention + embed / argmax", us_per_step=chain_s * 1e6, weight_bytes=w_bytes,
                                      achieved_GBs=w_bytes / chain_s / 1e9, frac=w_bytes / chain_s / 1e9 / HBM_PEAK_GBS))
    del eng
    want_cpu = world == 1 and not a.no_cpu_baseline and rank == 0
    cpu = cpu_baseline_decode(vitomr, lens, a.cpu_steps) if want_cpu else None
    del vitomr
    torch.cuda.empty_cache()

    # The headline is measured.  The secondary legs must never cost it: an exception inside one is caught (`_leg`), and a HANG (a collective that
    # one rank never reaches) is cut by a watchdog on every rank - after `--leg-timeout` seconds rank 0 prints the line with what it has and every
    # rank leaves the process.
    res = dict(mae=None, tf_step=None, ragged_decode=None, config5=None, latency_b1=None, end_to_end=None)
    res_lock = threading.Lock()
    printed = threading.Event()
    running = ["-"]

    def emit(timed_out):
        if printed.is_set():
            return
        printed.set()
        if rank != 0:
            return
        with res_lock:
            snap = dict(res)
        out = build_line(snap, timed_out)
        print(json.dumps(out), flush=True)

    def watchdog():
        # A leg that neither returns nor raises within --leg-timeout is a HANG (a collective one rank never reaches, a kernel that never drains).
        # The headline and the finished legs are still printed, but the process reports failure: stderr names the leg, exit code 3.
        if not printed.wait(a.leg_timeout):
            sys.stderr.write(f"bench.py: rank {rank}: leg '{running[0]}' did not finish within {a.leg_timeout} s - printing the partial line and exiting with code 3\n")
            sys.stderr.flush()
            emit(True)
            os._exit(3)

    def build_line(res, timed_out):
        mae_res, tf_res, rag_res, c5_res = res["mae"], res["tf_step"], res["ragged_decode"], res["config5"]
        out = dict(metric="LMX tokens/sec (greedy decode, KV cache)", value=tokens / dt, unit="tokens/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=dt / a.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                   config=dict(workload=f"vitomr_greedy_decode batch {a.batch}/GPU of {a.height}x{a.width} images ({S} patches), decode steps {a.warmup + 1}..{a.warmup + a.steps}" + ("" if a.warmup + a.steps <= cap else f" (re-armed every {cap} steps)"),
                               batch_per_gpu=a.batch, memory_len=S, decoder="12 x d1024 h16 mlp4096, V=227", hipgraph=use_graph,
                               split_merge=_split_merge_path()),
                   prefill_ms=prefill_s * 1e3, prefill_encoder_dtype=a.encoder_dtype, roofline=roof, cpu_baseline=cpu, mae=mae_res, tf_step=tf_res,
                   ragged_decode=rag_res, config5=c5_res, dist=dinfo,
                   value_steps256=(tokens / dt if a.steps == 256 else world * a.batch * 256 / dt256))
        if timed_out:
            out["legs_timed_out"] = f"secondary legs did not finish within {a.leg_timeout} s; fields still None were not measured"
        # decode-only (`value`) and end to end.  `end_to_end` = a TIMED call of inference() on the same batch shape (encoder + head +
        # prefill + 288 generated tokens per sequence, host polling included); `latency_b1` = BASELINE config 1 (one 256x1024 image) the same way.
        out["end_to_end"] = res["end_to_end"]
        out["latency_b1"] = res["latency_b1"]
        if cpu is not None:
            out["gpu_over_cpu"] = out["value"] / cpu["value"]
        return out

    threading.Thread(target=watchdog, daemon=True).start()

    def run_leg(key, fn, *args):
        running[0] = key
        r = _leg(fn, *args)
        with res_lock:
            res[key] = r

    if "e2e" in legs and world == 1:
        run_leg("end_to_end", bench_inference_call, dev, [(a.height, a.width)] * a.batch, 288, 3, False)
    if "b1" in legs and world == 1:
        run_leg("latency_b1", bench_inference_call, dev, [(256, 1024)], 256, 3, want_cpu)
    if "mae" in legs:
        run_leg("mae", bench_mae, dev, rank, world, dist, a.mae_batch, a.height, a.width, a.mae_steps, a.mae_dtype, want_cpu, a.grad_comm_dtype)
    if "tf" in legs and world == 1:
        run_leg("tf_step", bench_tf_step, dev, 16, a.height, a.width, 512, 4)
    if "ragged" in legs and world == 1:
        run_leg("ragged_decode", bench_ragged_decode, dev, 512)
    if "config5" in legs:
        run_leg("config5", bench_config5, dev, rank, world, dist, 32, 512, 2, a.grad_comm_dtype)
    running[0] = "-"
    emit(False)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
