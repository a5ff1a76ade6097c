"""Pin the CPU oracle against what the reference itself can execute: golden fixtures generated from
the REAL `/root/reference/utils/lora.py` + key converter (tests/golden/make_golden.py), and — when the reference is
mounted (build container) — direct execution of the reference code.  The UNet/VAE arithmetic lives in un-vendored
diffusers: that part is 'parity unpinned' by the reference and is cross-checked op-by-op against torch.nn.functional."""
import json
import os
import sys

import pytest
import torch
import torch.nn.functional as TF

from conftest import relerr

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF = "/root/reference"
SMALL = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=64)


def _gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _make(kind, st, r, mod):
    if kind.startswith("linear"):
        m = mod.LoraInjectedLinear(st["linear.weight"].shape[1], st["linear.weight"].shape[0], "linear.bias" in st, r=r)
    elif kind.startswith("conv2d"):
        w = st["conv.weight"]
        stride, pad = (2, 1) if kind == "conv2d_s2" else ((1, 0) if kind == "conv2d_1x1" else (1, 1))
        m = mod.LoraInjectedConv2d(w.shape[1], w.shape[0], w.shape[2], stride, pad, r=r)
    else:
        w = st["conv.weight"]
        m = mod.LoraInjectedConv3d(w.shape[1], w.shape[0], (3, 1, 1), (1, 0, 0), r=r)
    m.load_state_dict(st)
    return m.eval()


@pytest.mark.parametrize("flavour", ["oracle", "product"])
def test_lora_layers_reproduce_reference_outputs(flavour):
    if flavour == "oracle":
        import oracle.lora as mod
    else:
        import t2v_amd.utils.lora as mod
    gold = torch.load(os.path.join(GOLD, "lora_layers.pt"))
    for kind, it in gold.items():
        m = _make(kind, it["state"], it["r"], mod)
        with torch.no_grad():
            y = m(it["x"])
        assert torch.allclose(y, it["y"], atol=1e-5, rtol=1e-5), kind


@pytest.mark.parametrize("flavour", ["oracle", "product"])
def test_injection_matches_reference_injector(flavour):
    from oracle.unet3d import UNet3DConditionModel
    if flavour == "oracle":
        from oracle.lora import inject_trainable_lora_extended as inject
        model = UNet3DConditionModel(**SMALL)
    else:
        from t2v_amd.models.unet_3d_condition import UNet3DConditionModel as DUNet
        from t2v_amd.utils.lora import inject_trainable_lora_extended as inject
        model = DUNet(**SMALL)
    gold = _gold("lora_injection.json")
    params, names = inject(model, {"UNet3DConditionModel"}, r=4)
    wrapped = {n: type(m).__name__ for n, m in model.named_modules() if type(m).__name__.startswith("LoraInjected")}
    assert len(names) == gold["count"]
    assert sorted(wrapped) == sorted(gold["wrapped"])
    assert wrapped == gold["kinds"]
    assert sum(p.numel() for g in params for p in g) == gold["lora_params"]
    # restricted target list
    model2 = type(model)(**SMALL)
    _, names2 = inject(model2, {"Transformer2DModel", "ResnetBlock2D"}, r=4)
    assert len(names2) == gold["count_t2d_resnet"]


def test_full_model_facts():
    """1 411.2 M parameters, 574 LoRA-visible layers, diffusers key schema accepted by the reference's converter."""
    from oracle.unet3d import UNet3DConditionModel
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel as DUNet
    gold = _gold("unet_facts.json")
    for ctor in (UNet3DConditionModel, DUNet):
        with torch.device("meta"):
            m = ctor()
        assert sum(p.numel() for p in m.parameters()) == gold["n_params"] == 1411233860
        assert sum(1 for x in m.modules() if type(x) in (torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d)) == 574
        assert sorted(m.state_dict().keys()) == gold["diffusers_keys"]
    assert gold["unmapped_hf_keys"] == []          # every key went through convert_unet_state_dict's map
    assert gold["lora_size"] == {"4": 7321248, "16": 29246112, "32": 58479264}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted (GPU box)")
def test_reference_injector_on_dropin_model_and_live_layers():
    """Execute the REAL reference code: its injector on the drop-in model, its layers against the oracle's."""
    import contextlib
    import io
    sys.path.insert(0, REF)
    try:
        from utils import lora as ref_lora
    finally:
        sys.path.remove(REF)
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel as DUNet
    import oracle.lora as olora
    model = DUNet(**SMALL)
    base_ids = {id(p) for p in model.parameters()}
    with contextlib.redirect_stdout(io.StringIO()):
        params, names = ref_lora.inject_trainable_lora_extended(model, {"UNet3DConditionModel"}, r=4)
    assert len(names) == _gold("lora_injection.json")["count"]
    # wrappers share the base Parameters (no copy) — utils/lora.py:421,439,455
    shared = [p for n, p in model.named_parameters() if "lora_" not in n]
    assert all(id(p) in base_ids for p in shared)
    # name substrings train.py selects on (train.py:200,222,230,328-333)
    pnames = [n for n, _ in model.named_parameters()]
    for sub in ("attn1", "attn2", "temp_conv", ".attentions", "attn1.to_out", "lora", "temp"):
        assert any(sub in n for n in pnames), sub
    torch.manual_seed(3)
    a = ref_lora.LoraInjectedConv3d(16, 16, (3, 1, 1), (1, 0, 0), r=4).eval()
    b = olora.LoraInjectedConv3d(16, 16, (3, 1, 1), (1, 0, 0), r=4).eval()
    torch.nn.init.normal_(a.lora_up.weight, std=0.2)
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 16, 5, 4, 3)
    assert torch.allclose(a(x), b(x), atol=1e-6)


def test_oracle_leaf_ops_against_torch_functional():
    from oracle.unet3d import Attention, TemporalConvLayer, Timesteps
    torch.manual_seed(0)
    att = Attention(128, None, heads=2, dim_head=64).double()
    x = torch.randn(3, 10, 128, dtype=torch.float64)
    q, k, v = att.to_q(x), att.to_k(x), att.to_v(x)
    sp = lambda t: t.view(3, 10, 2, 64).transpose(1, 2)
    ref = TF.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(3, 10, 128)
    assert torch.allclose(att(x), att.to_out[0](ref), atol=1e-10)
    tc = TemporalConvLayer(32, 32).eval()
    torch.nn.init.normal_(tc.conv4[-1].weight, std=0.1)
    h = torch.randn(2 * 5, 32, 4, 4)
    y = tc(h, num_frames=5)
    assert y.shape == h.shape and not torch.allclose(y, h)
    # frame independence check: permuting pixels commutes with the (3,1,1) conv
    perm = torch.randperm(16)
    hp = h.flatten(2)[:, :, perm].view_as(h)
    assert torch.allclose(tc(hp, 5).flatten(2), y.flatten(2)[:, :, perm], atol=1e-5)
    te = Timesteps(320, True, 0)(torch.tensor([0, 999]))
    assert te.shape == (2, 320) and torch.allclose(te[0, :160], torch.ones(160)) and torch.allclose(te[0, 160:], torch.zeros(160))


def test_scheduler_matches_definition_and_zero_terminal_snr():
    from oracle import scheduler as S
    from t2v_amd.schedulers import DDPMScheduler
    acp = S.alphas_cumprod()
    d = DDPMScheduler()
    assert torch.allclose(d.alphas_cumprod, acp)
    x0, eps, t = torch.randn(2, 4, 3, 5, 5), torch.randn(2, 4, 3, 5, 5), torch.tensor([10, 900])
    assert torch.allclose(d.add_noise(x0, eps, t), S.add_noise(x0, eps, t))
    xt = S.add_noise(x0, eps, t)
    a = acp[t].view(2, 1, 1, 1, 1)
    assert torch.allclose((xt - (1 - a).sqrt() * eps) / a.sqrt(), x0, atol=1e-4)
    b2 = S.enforce_zero_terminal_snr(S.scaled_linear_betas())
    acp2 = torch.cumprod(1 - b2, 0)
    assert acp2[-1].abs() < 1e-8 and torch.allclose(acp2[0], acp[0], atol=1e-6)      # train.py:360-389 property


def test_fast_temporal_conv3d_is_the_same_convolution():
    """The CPU-baseline helper (Conv3d (k,1,1) evaluated as conv2d) must not change results."""
    from oracle.fastconv import fast_temporal_conv3d
    from oracle.unet3d import TemporalConvLayer
    torch.manual_seed(0)
    tc = TemporalConvLayer(64, 64).eval()
    torch.nn.init.normal_(tc.conv4[-1].weight, std=0.1)
    x = torch.randn(2 * 6, 64, 5, 7)
    y0 = tc(x, num_frames=6)
    with fast_temporal_conv3d():
        y1 = tc(x, num_frames=6)
    assert torch.allclose(y0, y1, atol=1e-5, rtol=1e-5)
    assert not hasattr(torch.nn.Conv3d, "_t2v_orig_conv_forward")


def test_product_zero_terminal_snr_matches_oracle_restatement():
    from oracle import scheduler as S
    from t2v_amd.schedulers import DDPMScheduler, enforce_zero_terminal_snr
    b = S.scaled_linear_betas()
    assert torch.allclose(enforce_zero_terminal_snr(b), S.enforce_zero_terminal_snr(b))
    d = DDPMScheduler()
    acp = d.alphas_cumprod.clone()
    d.rescale_betas()
    assert torch.equal(d.alphas_cumprod, acp)      # reference quirk: add_noise is unaffected (train.py:689-690)


def test_ms_webui_key_remap_matches_reference_converter():
    """utils/convert_diffusers_to_original_ms_text_to_video.py: every state-dict key of the drop-in UNet and of a stable_lora
    LoRA state dict maps to the name (and tensor rank) the REFERENCE converter produces (fixture generated by running it,
    tests/golden/make_golden_keymap.py)."""
    import json
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    from t2v_amd.stable_lora import lora as SL
    from t2v_amd.utils.convert_diffusers_to_original_ms_text_to_video import convert_unet_state_dict
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ms_keymap.json")))
    cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in gold["config"].items()}
    torch.manual_seed(0)
    unet = UNet3DConditionModel(**cfg)
    sd = unet.state_dict()
    out = convert_unet_state_dict(sd)
    assert list(sd.keys()) == list(gold["full"].keys())
    assert {k: [nk, list(out[nk].shape)] for k, nk in zip(sd.keys(), out.keys())} == gold["full"]
    SL.add_lora_to(unet, target_module=["Transformer2DModel", "ResnetBlock2D", "TransformerTemporalModel", "TemporalConvLayer"],
                   search_class=[torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d], r=4)()
    ld = SL.lora_state_dict(unet)
    lout = convert_unet_state_dict(ld, strict_mapping=True)
    assert {k: [nk, list(lout[nk].shape)] for k, nk in zip(ld.keys(), lout.keys())} == gold["stable_lora"]


def test_dropout_masks_of_consecutive_epochs_are_not_shifted_copies():
    """The epoch is hashed before it meets the seed (csrc/common.h eff_seed, oracle/dropout.effective_seed): with the first
    protocol (seed + epoch * G on the index lattice) the mask of step e+1 was the mask of step e shifted by one flat element.
    Consecutive epochs must agree with each other — under every small shift — only at the chance rate."""
    import torch
    from oracle.dropout import effective_seed, keep_mask
    rows, cols, p, seed = 64, 256, 0.3, 0x5EED1234
    n = rows * cols
    chance = p * p + (1 - p) * (1 - p)                       # P(two independent masks agree) = 0.58
    for e in (1, 2, (1 << 32) + 7):
        a = keep_mask(effective_seed(seed, e), rows, cols, p).flatten()
        b = keep_mask(effective_seed(seed, e + 1), rows, cols, p).flatten()
        assert abs(float(a.float().mean()) - (1 - p)) < 0.02
        for shift in range(-3, 4):
            lo, hi = max(0, shift), min(n, n + shift)
            agree = float((a[lo:hi] == b[lo - shift:hi - shift]).float().mean())
            assert abs(agree - chance) < 0.03, (e, shift, agree)
    # and different seeds under one epoch stay distinct masks
    assert not torch.equal(keep_mask(effective_seed(1, 5), 8, 64, p), keep_mask(effective_seed(2, 5), 8, 64, p))
