"""One secondary bench leg alone (for rocprofv3): python3 tools/prof_leg.py mae|tf|ragged|c5mae|c5tf   (c5*: config 5's RAGGED training steps,
32 images 256x1024 ... 768x3072, in one process without the gradient all-reduce)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
which = sys.argv[1]
if which == "mae":
    print(json.dumps(bench.bench_mae(dev, 0, 1, None, 32, 512, 2048, 3, "bf16", False)))
elif which == "tf":
    print(json.dumps(bench.bench_tf_step(dev, 16, 512, 2048, 512, 2)))
elif which in ("c5mae", "c5tf"):
    import time
    from torch.amp import autocast
    from acai_omr_amd.config import (ENCODER_FINE_TUNE_DEPTH, MASK_RATIO, MAX_LMX_SEQ_LEN, NUM_DECODER_LAYERS, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH)
    from acai_omr_amd.models.models import MAE, FineTuneOMREncoder, MAELoss, OMRCELoss, OMRDecoder, ScheduledSamplingViTOMR
    from acai_omr_amd.optim import FusedAdamW
    shapes = [bench.CONFIG4_SHAPES[i % 8] for i in range(32)]
    g = torch.Generator().manual_seed(3000)
    imgs = [torch.rand(1, *s, generator=g).to(dev) for s in shapes]
    torch.manual_seed(0)
    if which == "c5mae":
        model = MAE(MASK_RATIO, PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH).to(dev).train()
        opt = FusedAdamW(model.parameters(), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
        data = list(zip(imgs, imgs))

        def step():
            opt.zero_grad(set_to_none=True)
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                pred, loss_mask, target, _ = model.forward_packed(data)
            loss = MAELoss()(pred, loss_mask, target)
            loss.backward()
            opt.step()
    else:
        ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        enc = FineTuneOMREncoder(PATCH_SIZE, PE_MAX_HEIGHT, PE_MAX_WIDTH, ENCODER_FINE_TUNE_DEPTH, transformer_dropout=0.0)
        dec = OMRDecoder(MAX_LMX_SEQ_LEN, os.path.join(ROOT, "lmx_vocab.txt"), num_layers=NUM_DECODER_LAYERS, transformer_dropout=0.0)
        model = ScheduledSamplingViTOMR(enc, None, dec, transition_head_dropout=0.0).to(dev).train()
        groups, _ = model.create_fine_tune_param_groups(1e-4, 1e-5, 0.9)
        opt = FusedAdamW(groups, betas=(0.9, 0.95), weight_decay=0.01)
        loss_fn = OMRCELoss(dec.pad_idx)
        lmx = [torch.cat([torch.tensor([0]), torch.randint(3, 227, (512,), generator=g), torch.tensor([2])]).to(dev) for _ in imgs]
        data = list(zip(imgs, lmx))

        def step():
            opt.zero_grad(set_to_none=True)
            with autocast(device_type="cuda", dtype=torch.bfloat16):
                pred, tgt = model.forward_train(data, 0.7, 0.5, False)
            loss_fn(pred, tgt).backward()
            opt.step()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(json.dumps({"leg": which, "ms_per_step": (time.perf_counter() - t0) / 3 * 1e3, "images": len(imgs)}))
else:
    print(json.dumps(bench.bench_ragged_decode(dev, 512)))
