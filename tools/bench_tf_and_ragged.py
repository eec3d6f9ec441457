"""Measured side configs (GPU box):
  config 3 - ScheduledSamplingViTOMR.forward_train + OMRCELoss + backward, batch 16 x (512x2048 image, 512 LMX tokens), bf16 autocast
  config 4 - greedy decode of a ragged batch of 8 mixed-resolution systems (256x1024 ... 768x3072), hipGraph replay
Prints one JSON line per config."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.amp import autocast
from acai_omr_amd.config import ENCODER_FINE_TUNE_DEPTH, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH
from acai_omr_amd.models.models import FineTuneOMREncoder, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"

if which in ("tf", "both"):
    torch.manual_seed(0)
    enc = FineTuneOMREncoder(PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH, ENCODER_FINE_TUNE_DEPTH, transformer_dropout=0.0)
    dec = OMRDecoder(MAX_LMX_SEQ_LEN, os.path.join(ROOT, "lmx_vocab.txt"), num_layers=NUM_DECODER_LAYERS, transformer_dropout=0.0)
    m = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dropout=0.0).to(dev).train()
    g = torch.Generator().manual_seed(1)
    B, T = 16, 512
    batch = [(torch.rand(1, 512, 2048, generator=g).to(dev), torch.cat([torch.tensor([0]), torch.randint(3, 227, (T,), generator=g), torch.tensor([2])]).to(dev)) for _ in range(B)]
    loss_fn = OMRCELoss(dec.pad_idx)

    def step():
        m.zero_grad(set_to_none=True)
        with autocast(device_type="cuda", dtype=torch.bfloat16):
            pred, tgt = m.forward_train(batch, 0.7, 0.5, False)
        loss = loss_fn(pred, tgt)
        loss.backward()
        return loss
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 2
    for _ in range(n): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(json.dumps(dict(config="config 3: scheduled-sampling teacher-forced train step (two decoder passes), batch 16 x 512x2048, T=512, bf16 autocast, dropout 0",
                          ms_per_step=dt * 1e3, images_per_s=B / dt, tflops_algorithmic=7.14e12 * B / dt / 1e12, loss=float(loss.detach()))), flush=True)
    del m, enc, dec, batch
    torch.cuda.empty_cache()

if which in ("ragged", "both"):
    from acai_omr_amd.inference.vitomr_inference import set_up_omr_inference
    torch.manual_seed(0)
    vitomr, _ = set_up_omr_inference(os.path.join(ROOT, "lmx_vocab.txt"), max_batch_size=8, device=dev)
    vitomr.eval()
    shapes = [(256, 1024), (256, 2048), (384, 1536), (512, 2048), (512, 3072), (640, 2560), (768, 2304), (768, 3072)]
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(1, h, w, generator=g).to(dev) for h, w in shapes]
    with torch.no_grad():
        def prefill():
            lat32, _, lens = vitomr.encoder.forward_packed(imgs)
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                mem = vitomr.transition_head.forward_packed(lat32)
            vitomr.decoder.decoder_blocks.prepare_caches_packed(None, mem, lens)
            return lens
        prefill(); torch.cuda.synchronize()
        t0 = time.perf_counter(); lens = prefill(); torch.cuda.synchronize(); pf = time.perf_counter() - t0
    eng = vitomr.decoder.decoder_blocks.engine(dev)
    steps, warm = 512, 16
    cur = torch.cuda.current_stream(dev)
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        eng.arm(eng.B); eng.ensure_graph(1); eng.ensure_graph(eng.STEPS_PER_GRAPH); eng.arm(eng.B)
        eng.launch_steps(warm); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.launch_steps(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    step_bytes = 352.8e6 + sum(12 * 2 * (s + warm + steps // 2) * 1024 * 2 for s in lens)
    print(json.dumps(dict(config="config 4: greedy decode, ragged batch of 8 systems 256x1024..768x3072 (sum N = %d patches), 512 steps, hipGraph" % sum(lens),
                          tokens_per_s=8 * steps / dt, ms_per_step=dt / steps * 1e3, prefill_ms=pf * 1e3, step_GBs_algorithmic=step_bytes / (dt / steps) / 1e9, memory_lens=lens)), flush=True)
