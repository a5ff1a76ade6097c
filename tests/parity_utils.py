"""Shared helpers of the model-level parity tests (test infrastructure; imports oracle/ — never imported by the product).

Everything random is drawn on the HOST from fixed seeds (torch CPU generators), so the oracle results of the full-size
configurations can be computed once in the build container (tests/golden/make_oracle_step.py) and shipped as small
fixtures: the GPU box rebuilds the identical weights/inputs, checks their checksum against the fixture, and compares the
native path with the recorded oracle loss / LoRA-factor gradients.  If the checksum does not match (different torch
build), the tests fall back to running the oracle live.
"""
import json
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=64)
VAE_SMALL = dict(block_out_channels=(32, 64, 64, 64))
CONFIGS = {            # BASELINE.json configs: name -> (frames, H, W, lora rank)
    "c1": (8, 128, 128, 4),
    "c2": (16, 256, 256, 16),
    # configs[2] (full UNet finetune, train.py:172-236: every one of the 1.41 B UNet parameters trainable, no LoRA) — the
    # fixture is taken on the C1 clip: the weight-gradient layouts under test do not depend on the clip size, and the CPU
    # oracle's full-size backward stays at minutes / tens of GB
    "c3": (8, 128, 128, 0),
    # the same on configs[2]'s OWN clip (16 frames @256x256; round 4): one oracle pass, doubled (the passes are identical with a frozen
    # text encoder and no dropout) — ~10 min and ~45 GB on the build host
    "c3full": (16, 256, 256, 0),
}


def build_oracle(full=True, r=4, lora_up_scale=0.0, seed=0):
    """CPU fp32 oracle UNet (+ LoRA on every Linear/Conv, utils/lora.py:393-480) and VAE encoder, host-seeded."""
    from oracle.lora import inject_trainable_lora_extended
    from oracle.unet3d import UNet3DConditionModel
    from oracle.vae import AutoencoderKLEncoder
    from oracle.weights import randomize_lora_up, randomize_temporal_conv4
    torch.manual_seed(seed)
    unet = UNet3DConditionModel(**({} if full else SMALL))
    randomize_temporal_conv4(unet)
    vae = AutoencoderKLEncoder(**({} if full else VAE_SMALL)).eval()
    unet.requires_grad_(False)
    vae.requires_grad_(False)
    _, names = inject_trainable_lora_extended(unet, {"UNet3DConditionModel"}, r=r)
    randomize_lora_up(unet, scale=lora_up_scale)     # 0 => the reference's init (up = 0, utils/lora.py:55)
    for m in unet.modules():                         # eval_train mode (train.py:779-781): dropout off
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    unet.train()
    return unet, vae, len(names)


def build_oracle_full_finetune(full=True, seed=0):
    """CPU fp32 oracle UNet with EVERY parameter trainable (config C3) and the VAE encoder, host-seeded."""
    from oracle.unet3d import UNet3DConditionModel
    from oracle.vae import AutoencoderKLEncoder
    from oracle.weights import randomize_temporal_conv4
    torch.manual_seed(seed)
    unet = UNet3DConditionModel(**({} if full else SMALL))
    randomize_temporal_conv4(unet)
    vae = AutoencoderKLEncoder(**({} if full else VAE_SMALL)).eval()
    vae.requires_grad_(False)
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    unet.train()
    return unet, vae


def build_native_full_finetune(ounet, ovae, full=True):
    import t2v_amd  # noqa: F401
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    from t2v_amd.models.vae import AutoencoderKL
    with torch.device("meta"):
        dunet = UNet3DConditionModel(**({} if full else SMALL))
        dvae = AutoencoderKL(**({} if full else VAE_SMALL))
    dunet = dunet.to_empty(device="cuda")
    dvae = dvae.to_empty(device="cuda")
    dunet.load_state_dict(ounet.state_dict(), strict=True)
    dvae.load_state_dict(ovae.state_dict(), strict=True)
    dvae.requires_grad_(False)
    for m in dunet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    dunet.train()
    return dunet, dvae.eval()


def build_native(ounet, ovae, full=True, r=4):
    """The drop-in UNet/VAE on cuda:0 carrying exactly the oracle's weights (frozen + LoRA factors)."""
    import t2v_amd  # noqa: F401
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    from t2v_amd.models.vae import AutoencoderKL
    from t2v_amd.utils.lora import inject_trainable_lora_extended
    with torch.device("meta"):
        dunet = UNet3DConditionModel(**({} if full else SMALL))
        dvae = AutoencoderKL(**({} if full else VAE_SMALL))
    dunet = dunet.to_empty(device="cuda")
    dvae = dvae.to_empty(device="cuda")
    dunet.requires_grad_(False)
    dvae.requires_grad_(False)
    inject_trainable_lora_extended(dunet, {"UNet3DConditionModel"}, r=r)
    dunet.load_state_dict(ounet.state_dict(), strict=True)
    dvae.load_state_dict(ovae.state_dict(), strict=True)
    for m in dunet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    dunet.train()
    return dunet, dvae.eval()


DROPOUT_BASE_SEED = 0xD0C5


def enable_reference_dropout(net, p_lora=0.1, p_temporal=0.1):
    """Back to the reference constructors' dropout rates (LoRA wrappers: utils/lora.py:35,89; TemporalConvLayer:
    models/unet_3d_blocks.py:312) — build_oracle / build_native switch them off for the eval_train comparisons."""
    for _, m in net.named_modules():
        cls = m.__class__.__name__
        if cls in ("LoraInjectedLinear", "LoraInjectedConv2d", "LoraInjectedConv3d"):
            m.dropout.p = p_lora
        elif cls == "TemporalConvLayer":
            for seq in (m.conv2, m.conv3, m.conv4):
                for sub in seq:
                    if isinstance(sub, torch.nn.Dropout) or sub.__class__.__name__ == "ProtocolDropout":
                        sub.p = p_temporal
    net.train()
    return net


def weight_checksum(unet, vae):
    """Order-sensitive fp64 fingerprint of every parameter (frozen and LoRA) of both models."""
    acc, k = 0.0, 1
    for mod in (unet, vae):
        for _, p in sorted(mod.named_parameters()):
            v = p.detach().double().flatten()
            acc += float(v.sum()) * (1.0 + 1e-3 * (k % 97)) + float(v[:: max(1, v.numel() // 64)].abs().sum()) * 1e-2
            k += 1
    return acc


def oracle_loss_and_grads(unet, vae, batch, single_pass_doubled=False, sequential_passes=False):
    """eps-MSE of train.py:793-834 and its gradients w.r.t. every LoRA factor, on the CPU in fp32.
    `single_pass_doubled`: with a frozen text encoder the two passes of train.py:814-834 are identical computations, so
    L = 2 L0 and g = 2 g0 exactly; evaluating one pass halves the host memory of the full-size C2 run."""
    from oracle import scheduler
    from oracle.fastconv import fast_temporal_conv3d
    from oracle.train_step import finetune_unet_loss
    from oracle.vae import tensor_to_vae_latent
    for p in unet.parameters():
        p.grad = None
    with fast_temporal_conv3d():
        if single_pass_doubled:
            with torch.no_grad():
                latents = tensor_to_vae_latent(batch["pixel_values"], vae, batch["vae_eps"])
            noisy = scheduler.add_noise(latents, batch["noise"], batch["timesteps"], None)
            pred = unet(noisy, batch["timesteps"], encoder_hidden_states=batch["encoder_hidden_states"]).sample
            loss = 2.0 * torch.nn.functional.mse_loss(pred.float(), batch["noise"].float())
        elif sequential_passes:
            # the two passes of train.py:814-834 one after the other, each with its own backward (gradients add up in .grad): the
            # same loss and gradients as `(mse0 + mse1).backward()` with a frozen text encoder, at half the host memory — the
            # full-size C2 run WITH dropout (the passes draw different masks, so neither can stand in for the other)
            with torch.no_grad():
                latents = tensor_to_vae_latent(batch["pixel_values"], vae, batch["vae_eps"])
            noisy = scheduler.add_noise(latents, batch["noise"], batch["timesteps"], None)
            total = 0.0
            for _ in range(2):
                pred = unet(noisy, batch["timesteps"], encoder_hidden_states=batch["encoder_hidden_states"]).sample
                li = torch.nn.functional.mse_loss(pred.float(), batch["noise"].float())
                li.backward()
                total += float(li.detach())
                del pred, li
            grads = {n: p.grad.detach().clone() for n, p in unet.named_parameters() if p.requires_grad}
            return total, grads
        else:
            loss, _ = finetune_unet_loss(unet, vae, batch)
        loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in unet.named_parameters() if p.requires_grad}
    return float(loss.detach()), grads


def native_loss_and_grads(trainer, dunet, batch):
    """One forward+backward of the native trainer (fused LoRA path, flat gradient buffer); gradients are read back through
    each Parameter's `.grad` — a view of the flat buffer in the parameter's OWN layout, i.e. this un-permutes the
    lora_bank storage plan (GEMM-layout down factors, transposed / block-diagonal up factors)."""
    trainer.opt.zero_grad()
    loss = trainer._fwd_bwd({k: v.cuda() for k, v in batch.items()})
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in dunet.named_parameters() if p.requires_grad}
    return float(loss), grads


def compare_grads(g_ref, g_dut, share=1e-3):
    """(relative error of the whole gradient vector, its cosine, worst per-tensor relative error and worst per-tensor
    cosine among tensors holding >= `share` of the gradient norm, number of such tensors)."""
    num = den = dot = dd = 0.0
    for n, a in g_ref.items():
        a = a.double().flatten()
        b = g_dut[n].double().flatten()
        num += float((b - a).pow(2).sum()); den += float(a.pow(2).sum()); dot += float((a * b).sum()); dd += float(b.pow(2).sum())
    worst_rel, worst_cos, counted, per = 0.0, 1.0, 0, []
    for n, a in g_ref.items():
        a = a.double().flatten()
        b = g_dut[n].double().flatten()
        na = float(a.pow(2).sum())
        if na >= share * share * den and na > 0:
            counted += 1
            r = (float((b - a).pow(2).sum()) / na) ** 0.5
            c = float((a * b).sum()) / (na ** 0.5 * max(float(b.pow(2).sum()) ** 0.5, 1e-300))
            per.append((round(c, 4), round(r, 4), n))
            worst_rel, worst_cos = max(worst_rel, r), min(worst_cos, c)
    per.sort()
    return dict(rel=(num / max(den, 1e-300)) ** 0.5, cos=dot / max((den * dd) ** 0.5, 1e-300), worst_rel=worst_rel,
                worst_cos=worst_cos, tensors=counted, worst=per[:6])


def fixture_path(config, scale, dropout=False):
    return os.path.join(GOLDEN, f"oracle_step_{config}_s{scale:g}{'_drop' if dropout else ''}.pt")


def sampled_names(names, every=12):
    return sorted(names)[::every]


# ---------------------------------------------------------------------------------------------- shared by the model-level parity tests
FLOOR_FACTOR = 1.6          # a native result is accepted within 1.6x the error of the reference's own bf16 recipe (autocast floor);
                            # round 6: 2.0 -> 1.6 (worst ratio measured over rounds 3-5: 1.36 per tensor, 1.29 whole gradient)
RESULTS = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "parity_r06.jsonl")


def floor_row(config, scale):
    """Row of tests/golden/autocast_floor_<config>.json (scripts/autocast_floor.py) nearest to the `lora_up` amplitude."""
    with open(os.path.join(GOLDEN, f"autocast_floor_{config}.json")) as f:
        rows = json.load(f)["rows"]
    return min(rows, key=lambda r: abs(r["lora_up_scale"] - scale))


def record(**kw):
    """Append one result row to gpurun_out/parity_r05.jsonl (copied to profiles/ after a GPU run) and print it."""
    try:
        os.makedirs(os.path.dirname(RESULTS), exist_ok=True)
        with open(RESULTS, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass
    print(json.dumps(kw))


def set_lora_up(ounet, dunet, scale):
    from oracle.weights import randomize_lora_up
    randomize_lora_up(ounet, scale=scale)
    od = dict(ounet.named_parameters())
    with torch.no_grad():
        for n, p in dunet.named_parameters():
            if p.requires_grad:
                p.copy_(od[n])          # p.data is a view of the trainer's flat fp32 buffer
