"""Full-size parity (GPU): ModelScope-1.7B shapes, config C1 (8 frames @128x128, LoRA r=4, batch 1) — the
north-star bar: native eps-MSE within 1e-3 relative of the CPU fp32 oracle for identical seeds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_full_model_c1_loss_parity():
    import t2v_amd  # noqa: F401
    from oracle.lora import inject_trainable_lora_extended as oinject
    from oracle.train_step import finetune_unet_loss
    from oracle.unet3d import UNet3DConditionModel as OUNet
    from oracle.vae import AutoencoderKLEncoder
    from oracle.weights import randomize_lora_up, randomize_temporal_conv4, synthetic_batch
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    from t2v_amd.models.vae import AutoencoderKL
    from t2v_amd.training import DenoiseTrainer
    from t2v_amd.utils.lora import inject_trainable_lora_extended as dinject
    torch.manual_seed(0)
    with torch.device("cuda"):                    # random init on the GPU is ~50x faster than on the host
        ounet = OUNet()
        ovae = AutoencoderKLEncoder()
    ounet, ovae = ounet.cpu(), ovae.cpu().eval()
    randomize_temporal_conv4(ounet)
    ounet.requires_grad_(False)
    with torch.device("meta"):
        dunet = UNet3DConditionModel()
        dvae = AutoencoderKL()
    dunet = dunet.to_empty(device="cuda"); dvae = dvae.to_empty(device="cuda")
    dunet.load_state_dict(ounet.state_dict()); dvae.load_state_dict(ovae.state_dict())
    dunet.requires_grad_(False)
    _, names = oinject(ounet, {"UNet3DConditionModel"}, r=4)
    assert len(names) == 574                      # SURVEY.md §3.2 / tests/golden/unet_facts.json
    randomize_lora_up(ounet, scale=0.02)          # small non-zero up factors: LoRA branch live, net stays in its init regime
    dinject(dunet, {"UNet3DConditionModel"}, r=4)
    dunet.load_state_dict(ounet.state_dict(), strict=True)
    for m in list(ounet.modules()) + list(dunet.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ounet.train(); dunet.train()
    batch = synthetic_batch(8, 128, 128, seed=1234)
    lo, lat_o = finetune_unet_loss(ounet, ovae, batch)
    trainer = DenoiseTrainer(dunet, dvae.eval(), [p for p in dunet.parameters() if p.requires_grad], lr=5e-6)
    ld = trainer.loss_fn({k: v.cuda() for k, v in batch.items()})
    rel = abs(ld.item() - lo.item()) / abs(lo.item())
    print(f"C1 full model: loss oracle {lo.item():.6f} native {ld.item():.6f} rel {rel:.3e}")
    assert rel < 1e-3
