"""Host-side logic of the drop-in (CPU, no kernels): module-tree contract, LoRA handler, serialization, errors."""
import copy
import os

import pytest
import torch

SMALL = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=64)


def _unet():
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    torch.manual_seed(0)
    return UNet3DConditionModel(**SMALL)


def test_module_tree_contract():
    m = _unet()
    names = {type(x).__name__ for x in m.modules()}
    for cls in ("UNet3DConditionModel", "ResnetBlock2D", "TransformerTemporalModel", "Transformer2DModel", "Attention", "GEGLU",
                "TemporalConvLayer", "BasicTransformerBlock", "CrossAttnDownBlock3D", "DownBlock3D", "UNetMidBlock3DCrossAttn",
                "CrossAttnUpBlock3D", "UpBlock3D"):
        assert cls in names, cls
    for blk in list(m.down_blocks) + list(m.up_blocks) + [m.mid_block]:
        assert hasattr(blk, "gradient_checkpointing")
    assert m.mid_block.has_cross_attention and m.down_blocks[0].has_cross_attention and not hasattr(m.down_blocks[3], "has_cross_attention")
    tc = m.down_blocks[0].temp_convs[0].conv1[-1]
    assert type(tc) is torch.nn.Conv3d and tc.kernel_size == (3, 1, 1) and tc.padding == (1, 0, 0)
    blk = m.down_blocks[0].attentions[0].transformer_blocks[0]
    blk.attn1.set_processor(object()); blk.attn2.set_processor(object())          # train.py:138-150 seam
    m._set_gradient_checkpointing(value=True)
    assert m.mid_block.gradient_checkpointing and m.up_blocks[1].gradient_checkpointing
    assert m.config.in_channels == 4 and m.dtype == torch.float32


def test_cpu_forward_raises_no_silent_fallback():
    m = _unet()
    with pytest.raises(RuntimeError, match="ROCm device"):
        m(torch.zeros(1, 4, 2, 8, 8), torch.tensor([1]), torch.zeros(1, 77, 64))


def test_lora_handler_contract(tmp_path):
    from t2v_amd.utils.lora_handler import LoraHandler
    from t2v_amd.utils import lora as L
    m = _unet()
    m.requires_grad_(False)
    h = LoraHandler(use_unet_lora=True, unet_replace_modules=["UNet3DConditionModel"])
    params, negation = h.add_lora_to_model(True, m, h.unet_replace_modules, 0.1, None, r=4)
    import itertools
    plist = list(itertools.chain(*params))                   # train.py:212-218 consumes it exactly like this
    assert len(plist) == 2 * 573 and all(p.requires_grad for p in plist)
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert all("lora" in n for n in trainable)
    w = m.down_blocks[0].resnets[0].conv1
    assert isinstance(w, L.LoraInjectedConv2d) and w.conv.weight.requires_grad is False
    # CPU composition semantic: zero-init up => wrapper == base
    x = torch.randn(1, 64, 8, 8)
    assert torch.allclose(w.eval()(x), w.conv(x))
    # save -> fresh model -> load round trip (list-of-tensors .pt format, utils/lora.py:570-582)
    for mod in m.modules():
        if isinstance(mod, L._WRAPPERS):
            torch.nn.init.normal_(mod.lora_up.weight, std=0.05)
    h.save_lora_weights(m, str(tmp_path), step=7)
    f = tmp_path / "lora" / "7_unet.pt"                      # utils/lora_handler.py:336: `{save_path}/lora/{step}_unet.pt`
    assert f.exists() and not (tmp_path / "lora" / "7_text_encoder.pt").exists()
    m2 = _unet()
    h2 = LoraHandler(use_unet_lora=True)
    h2.add_lora_to_model(True, m2, ["UNet3DConditionModel"], 0.0, str(tmp_path / "lora"), r=4)
    a = dict(m.named_parameters()); b = dict(m2.named_parameters())
    assert all(torch.equal(a[n], b[n]) for n in a if "lora" in n)
    # collapse + remove restores a plain module tree with merged weights
    w0 = m.down_blocks[0].resnets[0].conv1
    delta = (w0.lora_up.weight.flatten(1) @ w0.lora_down.weight.flatten(1)).reshape(w0.conv.weight.shape)
    before = w0.conv.weight.detach().clone()
    L.collapse_lora(m, {"UNet3DConditionModel"})
    L.monkeypatch_remove_lora(m)
    c1 = m.down_blocks[0].resnets[0].conv1
    assert type(c1) is torch.nn.Conv2d and torch.allclose(c1.weight, before + delta, atol=1e-6)


def test_save_pretrained_roundtrip_and_deepcopy(tmp_path):
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    m = _unet()
    m.save_pretrained(str(tmp_path / "unet"))
    m2 = UNet3DConditionModel.from_pretrained(str(tmp_path), subfolder="unet")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    m3 = copy.deepcopy(m).cpu().to(torch.float32)
    assert sorted(m3.state_dict()) == sorted(m.state_dict())


def test_weight_preparation_layouts():
    """Prepared GEMM layouts (host logic, runs on CPU): fwd = [N, (tap, c)], bwd = [Cin, (flipped tap, N)]."""
    import t2v_amd.functional as F
    w = torch.randn(5, 3, 3, 3)
    f = F._prep_compute(w, "fwd", None).float()
    assert f.shape == (8, 9 * 8)
    assert torch.allclose(f.view(8, 3, 3, 8)[:5, 1, 2, :3], w[:, :, 1, 2].to(torch.bfloat16).float())
    b = F._prep_compute(w, "bwd", None).float()
    assert b.shape == (8, 9 * 8)
    assert torch.allclose(b.view(8, 3, 3, 8)[:3, 0, 1, :5], w[:, :, 2, 1].t().to(torch.bfloat16).float())
    g = torch.randn(8, 9 * 8)
    back = F._unprep_weight_grad(g, w, F.ConvCfg.conv2d(1, 4, 4))
    assert back.shape == w.shape and torch.equal(back[2, 1, 0, 2], g.view(8, 3, 3, 8)[2, 0, 2, 1])
    cfg = F.ConvCfg.conv2d(2, 8, 8, 3, 2, 1)
    assert (cfg.Ho, cfg.Wo) == (4, 4)
    bg = cfg.bwd_geom(16)
    assert (bg.Hv, bg.Wv, bg.Ho, bg.Wo, bg.py, bg.tdiv) == (4, 4, 8, 8, 1, 2)


def test_lora_bank_layouts_are_gemm_layouts():
    """lora_bank: the flat-buffer storage of every LoRA factor IS the layout the GEMM kernels consume
    (down: [rp, taps, Cin_p] with K ordered (tap, c); up: [rp, Np]) while the Parameter keeps the module's shape."""
    import t2v_amd.lora_bank as lb
    from t2v_amd.utils import lora as L
    torch.manual_seed(0)
    mods = torch.nn.ModuleDict(dict(
        lin=L.LoraInjectedLinear(24, 40, bias=True, r=4),
        c2d=L.LoraInjectedConv2d(16, 24, 3, 1, 1, r=4),
        c3d=L.LoraInjectedConv3d(16, 16, (3, 1, 1), (1, 0, 0), r=16)))
    for m in mods.values():
        torch.nn.init.normal_(m.lora_up.weight)
    plans = lb.plan(mods)
    assert len(plans) == 6
    for name, m in mods.items():
        for role, p in (("down", m.lora_down.weight), ("up", m.lora_up.weight)):
            e, r, _ = plans[id(p)]
            assert r == role
            n = e.down_numel if role == "down" else e.up_numel
            flat = torch.zeros(n)
            view = lb.param_view(flat, p, e, role)
            assert view.shape == p.shape
            view.copy_(p.detach())
            if role == "down":
                f3 = flat.view(e.rp, e.taps, e.cin_p)
                ref = p.detach().flatten(2).permute(0, 2, 1) if p.dim() > 2 else p.detach()[:, None, :]
                assert torch.equal(f3[: e.r, :, : e.cin], ref) and f3[e.r:].abs().sum() == 0
            else:
                f2 = flat.view(e.rp, e.npad)
                assert torch.equal(f2[: e.r, : e.n], p.detach().flatten(1).t()) and f2[e.r:].abs().sum() == 0


def test_stable_lora_flavour_cpu_semantics(tmp_path):
    """stable_lora mirror: layers re-materialise W + (B@A).view()*scaling (stable_lora/lora.py:119-126,190-197), injection
    shares weight/bias, only lora_ params train, full-weights safetensors round trip."""
    import torch.nn.functional as TF
    from t2v_amd.stable_lora import lora as SL
    from t2v_amd.utils.lora_handler import LoraHandler, LoraVersions
    torch.manual_seed(0)
    c = SL.Conv2d(8, 12, 3, r=4, lora_alpha=4, merge_weights=False, padding=1)
    torch.nn.init.normal_(c.lora_B, std=0.1)
    x = torch.randn(2, 8, 5, 5)
    w = c.weight + (c.lora_B @ c.lora_A).view(c.weight.shape) * 1.0
    assert torch.allclose(c(x), TF.conv2d(x, w, c.bias, padding=1), atol=1e-6)
    c3 = SL.Conv3d(8, 8, 3, r=4, lora_alpha=4, merge_weights=False, padding=(1, 0, 0))
    torch.nn.init.normal_(c3.lora_B, std=0.1)
    x3 = torch.randn(1, 8, 4, 3, 3)
    w3 = c3.weight + torch.mean((c3.lora_B @ c3.lora_A).view(8, 8, 3, 3, 1), dim=-2, keepdim=True)
    assert torch.allclose(c3(x3), TF.conv3d(x3, w3, c3.bias, padding=(1, 0, 0)), atol=1e-6)
    lin = SL.Linear(16, 8, r=4, lora_alpha=4, merge_weights=False)
    torch.nn.init.normal_(lin.lora_B, std=0.1)
    xl = torch.randn(3, 16)
    assert torch.allclose(lin(xl), TF.linear(xl, lin.weight + lin.lora_B @ lin.lora_A, lin.bias), atol=1e-6)
    m = _unet()
    h = LoraHandler(version=LoraVersions.stable_lora, use_unet_lora=True, save_for_webui=True)
    params, neg = h.add_lora_to_model(True, m, ["Transformer2DModel", "ResnetBlock2D"], 0.0, None, r=4)
    assert params is m and neg is None
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    assert trainable and all("lora_" in n for n in trainable)
    n_lora = sum(1 for x_ in m.modules() if isinstance(x_, SL._LORA_TYPES))
    assert n_lora == 271            # same layer count the reference cloneofsimo injector finds for this target list (golden)
    h.save_lora_weights(m, str(tmp_path), step=3)
    f = tmp_path / "lora" / "full_weights" / "3_lora_text_to_video_unet.safetensors"
    assert f.exists()
    from safetensors.torch import load_file
    web = load_file(str(tmp_path / "lora" / "webui_3_lora_text_to_video.safetensors"))      # ModelScope key layout, fp16
    assert len(web) == len(SL.lora_state_dict(m)) and all(v.dtype == torch.float16 for v in web.values())
    assert "input_blocks.1.1.transformer_blocks.0.attn1.to_q.lora_A" in web and "input_blocks.1.0.in_layers.2.lora_B" in web
    for x_ in m.modules():
        if isinstance(x_, SL._LORA_TYPES):
            torch.nn.init.normal_(x_.lora_B, std=0.02)
    before = {k: v.clone() for k, v in SL.lora_state_dict(m).items()}
    SL.load_lora(m, str(f))
    after = SL.lora_state_dict(m)
    assert any(not torch.equal(before[k], after[k]) for k in before)      # saved zeros for lora_B were restored


def test_lora_bank_projection_groups():
    """to_q/to_k/to_v sharing an input are stored as one group: downs back to back ([3rp, Cin]), ups as the block-diagonal
    [3rp, 3N] matrix (member i = row block i, column block i; the rest structurally zero), Parameters still module-shaped."""
    import t2v_amd.lora_bank as lb
    from t2v_amd.utils import lora as L

    class Attn(torch.nn.Module):
        def __init__(self, dim, kv):
            super().__init__()
            self.to_q = L.LoraInjectedLinear(dim, dim, bias=False, r=4)
            self.to_k = L.LoraInjectedLinear(kv, dim, bias=False, r=4)
            self.to_v = L.LoraInjectedLinear(kv, dim, bias=False, r=4)

    torch.manual_seed(0)
    model = torch.nn.ModuleDict(dict(selfa=Attn(24, 24), cross=Attn(24, 40), other=L.LoraInjectedLinear(24, 8, r=4)))
    for m in model.modules():
        if hasattr(m, "lora_up"):
            torch.nn.init.normal_(m.lora_up.weight)
    params = [p for n, p in model.named_parameters() if "lora_" in n]
    plans = lb.plan(model)
    gs, gc = model["selfa"]._t2v_group, model["cross"]._t2v_group
    assert gs.n == 3 and gc.n == 2 and gc.mods[0] is model["cross"].to_k
    assert (gs.rp, gs.npad, gs.cin_p) == (24, 72, 24) and (gc.rp, gc.npad, gc.cin_p) == (16, 48, 40)
    plist = lb.reorder(params, plans)
    assert len(plist) == len(params) and {id(p) for p in plist} == {id(p) for p in params}
    sizes = []
    for p in plist:
        e, role, _ = plans[id(p)]
        sizes.append(((e.down_numel if role == "down" else e.up_numel) + 7) // 8 * 8)
    flat, flat16, flatg = torch.zeros(sum(sizes)), torch.zeros(sum(sizes), dtype=torch.bfloat16), torch.zeros(sum(sizes))
    off, offsets = 0, {}
    for p, k in zip(plist, sizes):
        e, role, _ = plans[id(p)]
        v = lb.param_view(flat[off:off + k], p, e, role)
        assert v.shape == p.shape
        v.copy_(p.detach())
        offsets[id(p)] = off
        off += k
    flat16.copy_(flat)
    lb.attach(plans, flat16, flatg, offsets)
    for g in (gs, gc):
        n, rp, npad = g.n, g.rp_each, g.npad_each
        dcat = torch.cat([torch.nn.functional.pad(m.lora_down.weight.detach(), (0, 0, 0, rp - 4)) for m in g.mods], 0)
        assert torch.equal(g.down_w16.float(), dcat.bfloat16().float())
        ublk = torch.zeros(g.rp, g.npad)
        for i, m in enumerate(g.mods):
            ublk[i * rp: i * rp + 4, i * npad: i * npad + 24] = m.lora_up.weight.detach().t()
        assert torch.equal(g.up_w16.float(), ublk.bfloat16().float())
        for i, m in enumerate(g.mods):       # the per-member views used by the unfused path address the same storage
            e = m._t2v_bank
            assert e.up_w16.shape == (rp, npad) and e.up_w16.stride(0) == g.npad
            assert torch.equal(e.up_w16.float(), ublk[i * rp:(i + 1) * rp, i * npad:(i + 1) * npad].bfloat16().float())
            assert e.up_g.data_ptr() == g.up_g.data_ptr() + (i * rp * g.npad + i * npad) * 4
    # a partially trained group falls apart into ordinary layers
    plans2 = lb.plan(model)
    keep = [p for p in params if p is not model["selfa"].to_k.lora_up.weight]
    lb.reorder(keep, plans2)
    assert "_t2v_group" not in model["selfa"].__dict__ and plans2[id(model["selfa"].to_q.lora_up.weight)][0].group is None


def test_latent_cache_file_format(tmp_path):
    """handle_cache_latents / CachedDataset (train.py:266-314, utils/dataset.py:589-603): one `cached_{i}.pt` per batch, the
    'pixel_values' entry replaced by scaled latents (B,4,F,h,w) -> batch dim stripped, prompts kept, files listed sorted."""
    from t2v_amd.utils.latent_cache import CachedDataset, handle_cache_latents

    class _Dist:
        def __init__(self, x):
            self.x = x

        def sample(self, eps=None):
            return self.x[:, :4, ::8, ::8] * 2.0            # stand-in for the posterior sample at 1/8 resolution

    class _Vae:
        def encode(self, x):
            class R:
                latent_dist = _Dist(torch.cat([x, x[:, :1]], 1))
            return R

    batches = [{"pixel_values": torch.randn(1, 3, 3, 16, 16), "prompt_ids": torch.arange(77)[None, None], "text_prompt": [f"p{i}"]}
               for i in range(3)]
    assert handle_cache_latents(False, str(tmp_path), batches, 1, _Vae()) is None
    dl = handle_cache_latents(True, str(tmp_path), [dict(b) for b in batches], 1, _Vae())
    files = sorted(os.listdir(tmp_path / "cached_latents"))
    assert files == ["cached_0.pt", "cached_1.pt", "cached_2.pt"]
    ds = CachedDataset(str(tmp_path / "cached_latents"), map_location="cpu")
    item = ds[1]
    assert item["pixel_values"].shape == (4, 3, 2, 2) and item["prompt_ids"].shape == (1, 77) and item["text_prompt"] == "p1"
    x = batches[1]["pixel_values"]
    ref = (torch.cat([x[0], x[0][:, :1]], 1)[:, :4, ::8, ::8] * 2.0).permute(1, 0, 2, 3) * 0.18215
    assert torch.allclose(item["pixel_values"], ref)
    got = next(iter(torch.utils.data.DataLoader(ds, batch_size=1)))       # the loader the train loop consumes
    assert got["pixel_values"].shape == (1, 4, 3, 2, 2)
    dl2 = handle_cache_latents(True, str(tmp_path), None, 1, None, cached_latent_dir=str(tmp_path / "cached_latents"))
    assert len(dl2.dataset) == 3 and len(dl.dataset) == 3


def test_flat_buffer_rehoming_after_device_round_trip():
    """lora_bank.is_homed / rehome: after `module.cpu()`-style re-allocation (the reference's save_pipe, train.py:417-442) the
    Parameters are pointed back at their flat-buffer views with their CURRENT values; foreign gradients are merged, not lost."""
    import t2v_amd.lora_bank as lb
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 4)
    params = list(lin.parameters())
    flat_p, flat_g = torch.zeros(64), torch.zeros(64)
    homes, off = [], 0
    for p in params:
        v, g = flat_p[off:off + p.numel()].view(p.shape), flat_g[off:off + p.numel()].view(p.shape)
        v.copy_(p.detach()); p.data = v; p.grad = g
        homes.append((p, v, g.detach())); off += p.numel()          # as FlatAdamW does
    assert lb.is_homed(homes)
    lin._apply(lambda t: t.clone())                 # what .cpu()/.to(device) do: every parameter (and grad) gets a new storage
    assert not lb.is_homed(homes)
    with torch.no_grad():
        lin.weight.add_(1.0)                        # value changed while detached
    lin.weight.grad = torch.full_like(lin.weight, 0.5)        # a gradient that landed outside the flat buffer
    lb.rehome(homes)
    assert lb.is_homed(homes)
    assert lin.weight.data_ptr() == homes[0][1].data_ptr() and lin.weight.grad.data_ptr() == homes[0][2].data_ptr()
    assert torch.equal(flat_p[:24].view(4, 6), lin.weight.detach()) and torch.all(flat_g[:24] == 0.5)
    flat_p[:24] += 1.0                              # an optimizer update through the flat buffer is visible in the module
    assert torch.equal(lin.weight.detach(), flat_p[:24].view(4, 6))
    lin.zero_grad(set_to_none=True)                 # torch's default: grads become None -> re-attached, nothing to merge
    assert not lb.is_homed(homes)
    lb.rehome(homes)
    assert lin.weight.grad.data_ptr() == homes[0][2].data_ptr()


@pytest.mark.parametrize("kind,steps", [("ddim", 7), ("dpm", 7), ("dpm", 20)])
def test_sampling_schedulers_follow_the_exact_trajectory(kind, steps):
    """With an exact epsilon model (x0 known) both samplers must stay on x_t = alpha_t x0 + sigma_t eps and end at x0:
    DDIM by construction, DPM-Solver++(2M) because its update is exact for a constant data prediction."""
    from t2v_amd.schedulers import DDIMScheduler, DPMSolverMultistepScheduler
    sch = DDIMScheduler() if kind == "ddim" else DPMSolverMultistepScheduler()
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 3, 4, 4, generator=g, dtype=torch.float64)
    eps = torch.randn(1, 4, 3, 4, 4, generator=g, dtype=torch.float64)
    ts = sch.set_timesteps(steps)
    assert len(ts) == steps and int(ts[0]) == 999 and all(int(a) > int(b) for a, b in zip(ts, ts[1:]))
    acp = sch.alphas_cumprod.double()
    x = acp[ts[0]].sqrt() * x0 + (1 - acp[ts[0]]).sqrt() * eps
    for i, t in enumerate(ts):
        model_eps = (x - acp[t].sqrt() * x0) / (1 - acp[t]).sqrt()           # what a perfect model predicts at (x, t)
        x = sch.step(model_eps, t, x)
        if i + 1 < len(ts):
            tn = ts[i + 1]
            assert torch.allclose(x, acp[tn].sqrt() * x0 + (1 - acp[tn]).sqrt() * eps, atol=1e-9)
    assert torch.allclose(x, x0, atol=1e-9)


def test_sampler_cfg_loop_with_a_stub_unet():
    """TextToVideoSampler: CFG doubles the batch (uncond | cond), guidance mixes the two halves, the scheduler is stepped once
    per timestep; with a model that returns the true epsilon of a known x0 the loop lands on x0 for any guidance scale."""
    from t2v_amd.pipelines import TextToVideoSampler
    from t2v_amd.schedulers import DPMSolverMultistepScheduler
    sch = DPMSolverMultistepScheduler()
    x0 = torch.randn(1, 4, 2, 4, 4, generator=torch.Generator().manual_seed(1))
    calls = []

    class Stub:
        config = type("c", (), {"in_channels": 4})()

        def __call__(self, x, t, encoder_hidden_states=None):
            calls.append((x.shape[0], int(t[0]), encoder_hidden_states.shape[0]))
            acp = sch.alphas_cumprod[int(t[0])]
            return type("o", (), {"sample": (x - acp.sqrt() * x0) / (1 - acp).sqrt()})()

    pe, ne = torch.randn(1, 77, 8), torch.zeros(1, 77, 8)
    out = TextToVideoSampler(Stub(), sch)(pe, ne, num_frames=2, height=32, width=32, num_inference_steps=6, guidance_scale=7.5,
                                          generator=torch.Generator().manual_seed(2))
    assert out.shape == (1, 4, 2, 4, 4) and torch.allclose(out, x0, atol=1e-4)
    assert len(calls) == 6 and all(c[0] == 2 and c[2] == 2 for c in calls) and calls[0][1] == 999


def test_aborted_backward_leaves_no_stale_factor_gradient_launches():
    """The reference's loop swallows exceptions raised inside a step (train.py:881-883).  Factor-gradient descriptors queued by a
    backward pass that died half-way point at operands that may be freed: the next zero_grad() drops them (with a warning)
    instead of letting the next pass flush them."""
    import warnings
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    w = nv.LoraWgrad()
    w.rows, w.N, w.C = 10, 8, 8
    F._wq["descs"].append(w)
    F._wq["keep"].append(("operands",))
    F._wq["bytes"] = 320
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert F.drop_pending_wgrads() == 1
    assert len(rec) == 1 and "aborted backward" in str(rec[0].message)
    assert not F._wq["descs"] and not F._wq["keep"] and F._wq["bytes"] == 0
    assert F.drop_pending_wgrads() == 0


@pytest.mark.parametrize("kind,ws,rotate", [("dpm", 3, False), ("dpm", 4, True), ("ddim", 2, True), ("dpm", 16, True)])
def test_windowed_sampling_keeps_per_frame_solver_history(kind, ws, rotate):
    """`diffuse` of inference.py:153-267: windows of `window_size` frames per UNet call, the multistep history kept per frame by
    the caller, optional roll of the frame axis by a prime shift per timestep.  With a model that treats frames independently
    (but is nonlinear and time dependent, so the second-order history matters) the windowed / rotated loop must reproduce the
    whole-clip loop frame by frame."""
    from t2v_amd.pipelines import TextToVideoSampler, primes_up_to
    from t2v_amd.schedulers import DDIMScheduler, DPMSolverMultistepScheduler
    assert primes_up_to(12) == [2, 3, 5, 7, 11] and primes_up_to(2) == [2, 3]
    mk = (lambda: DDIMScheduler()) if kind == "ddim" else (lambda: DPMSolverMultistepScheduler())
    sizes = []

    class Stub:
        config = type("c", (), {"in_channels": 4})()

        def __call__(self, x, t, encoder_hidden_states=None):
            sizes.append(x.shape[2])
            return type("o", (), {"sample": torch.tanh(x * (0.3 + int(t[0]) / 2000.0)) + 0.1 * x})()

    pe, ne = torch.randn(1, 77, 8), torch.zeros(1, 77, 8)
    lat = torch.randn(1, 4, 10, 4, 4, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    kw = dict(num_inference_steps=9, guidance_scale=1.0, latents=lat.clone())
    whole = TextToVideoSampler(Stub(), mk())(pe, ne, **kw)
    n_whole = len(sizes)
    win = TextToVideoSampler(Stub(), mk())(pe, ne, window_size=ws, rotate=rotate, generator=torch.Generator().manual_seed(4),
                                          **dict(kw, latents=lat.clone()))
    assert n_whole == 9 and set(sizes[:9]) == {10}
    expect = [min(ws, 10 - s) for s in range(0, 10, min(ws, 10))] * 9
    assert sizes[9:] == expect                                   # one UNet call per window and timestep, ragged last window
    assert win.shape == whole.shape and torch.allclose(win, whole, atol=1e-10)


def test_decode_latents_batches_frames_and_restores_the_clip_layout():
    """inference.py:125-140: latents `(b c f h w)` -> frames `(b f)` -> `vae.decode(z / scaling_factor).sample` in batches of
    `vae_batch_size` -> `(b c f H W)` fp32; and the sampler's `decode=True` returns that."""
    from t2v_amd.models.vae import decode_latents
    from t2v_amd.pipelines import TextToVideoSampler
    seen = []

    class StubVae:
        config = type("c", (), {"scaling_factor": 0.5})()

        def decode(self, z):
            seen.append(tuple(z.shape))
            up = z[:, :3].repeat_interleave(8, 2).repeat_interleave(8, 3)
            return type("o", (), {"sample": up.half()})()

    lat = torch.arange(2 * 4 * 5 * 2 * 2, dtype=torch.float32).reshape(2, 4, 5, 2, 2)
    px = decode_latents(lat, StubVae(), batch_size=4)
    assert seen == [(4, 4, 2, 2), (4, 4, 2, 2), (2, 4, 2, 2)]
    assert px.shape == (2, 3, 5, 16, 16) and px.dtype == torch.float32
    assert torch.equal(px[1, 2, 3, :8, :8], torch.full((8, 8), float(lat[1, 2, 3, 0, 0]) / 0.5))

    class StubUnet:
        config = type("c", (), {"in_channels": 4})()

        def __call__(self, x, t, encoder_hidden_states=None):
            return type("o", (), {"sample": torch.zeros_like(x)})()

    out = TextToVideoSampler(StubUnet(), None, StubVae())(torch.zeros(1, 77, 8), None, num_frames=3, height=16, width=16,
                                                          num_inference_steps=2, guidance_scale=1.0, decode=True,
                                                          generator=torch.Generator().manual_seed(0))
    assert out.shape == (1, 3, 3, 16, 16)
    with pytest.raises(RuntimeError):
        TextToVideoSampler(StubUnet())(torch.zeros(1, 77, 8), None, num_frames=1, height=16, width=16, num_inference_steps=1,
                                      guidance_scale=1.0, decode=True, generator=torch.Generator().manual_seed(0))


def test_vae_from_pretrained_accepts_a_full_diffusers_checkpoint(tmp_path):
    """`AutoencoderKL.from_pretrained(path, subfolder="vae")` (the call in the reference's load_primary_models,
    train.py:119-123) on a checkpoint in the stock diffusers layout: encoder + decoder + post_quant_conv, with the
    ModelScope-era attention names query/key/value/proj_attn.  The decoder half is built because the checkpoint carries it
    (`decode` serves inference.py:125-140); `with_decoder=False` loads the encoder half only (the train step)."""
    import json
    from safetensors.torch import save_file
    from t2v_amd.models.vae import AutoencoderKL
    cfg = dict(block_out_channels=(32, 64, 64, 64))
    torch.manual_seed(0)
    ref = AutoencoderKL(with_decoder=True, **cfg)
    sd = {}
    for k, v in ref.state_dict().items():
        for a, b in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
            if ".attentions." in k:
                k = k.replace(a, b)
        sd[k] = v.clone()
    assert any(".query." in k for k in sd) and any(k.startswith("decoder.up_blocks.0.upsamplers.0.conv") for k in sd)
    assert sum(k.startswith("decoder.up_blocks.") and ".resnets." in k and k.endswith("conv1.weight") for k in sd) == 12   # 4 x 3
    d = tmp_path / "vae"
    d.mkdir()
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    with open(d / "config.json", "w") as f:
        json.dump(dict(_class_name="AutoencoderKL", block_out_channels=list(cfg["block_out_channels"]), latent_channels=4), f)
    m = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    a, b = ref.state_dict(), m.state_dict()
    assert m.decoder is not None and set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    enc = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae", with_decoder=False)
    assert enc.decoder is None and not any(k.startswith(("decoder.", "post_quant_conv.")) for k in enc.state_dict())
    assert all(torch.equal(a[k], v) for k, v in enc.state_dict().items())
    with pytest.raises(RuntimeError):
        enc.decode(torch.zeros(1, 4, 4, 4))
    # an unknown encoder-side key still fails loudly (the load stays strict)
    sd["encoder.bogus.weight"] = torch.zeros(1)
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError):
        AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")


def test_ddim_leading_grid_of_the_modelscope_scheduler_config():
    """`timestep_spacing="leading"`, `steps_offset=1`, `set_alpha_to_one=False` (ModelScope scheduler_config.json): the grid is
    t_i = i*(T//n)+1, the step past the end uses abar_0, and an exact model stays on the trajectory."""
    from t2v_amd.schedulers import DDIMScheduler
    sch = DDIMScheduler(timestep_spacing="leading", steps_offset=1, set_alpha_to_one=False)
    ts = sch.set_timesteps(50)
    assert [int(t) for t in ts[:3]] == [981, 961, 941] and int(ts[-1]) == 1
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 2, 4, 4, generator=g, dtype=torch.float64)
    eps = torch.randn(1, 4, 2, 4, 4, generator=g, dtype=torch.float64)
    acp = sch.alphas_cumprod.double()
    x = acp[ts[0]].sqrt() * x0 + (1 - acp[ts[0]]).sqrt() * eps
    for t in ts:
        x = sch.step((x - acp[t].sqrt() * x0) / (1 - acp[t]).sqrt(), t, x)
    assert torch.allclose(x, acp[0].sqrt() * x0 + (1 - acp[0]).sqrt() * eps, atol=1e-9)   # final_alpha_cumprod = abar_0


def test_stable_lora_embedding_follows_loralib():
    """`create_lora_emb` (stable_lora/lora.py:241-248): CLIPTextEmbeddings' tables get loralib's Embedding — E[x] + (A^T[x] B^T) alpha/r,
    A zeros / B normal, base table shared and frozen, `lora_` keys in the state dict (what reference checkpoints carry)."""
    import torch
    from t2v_amd.stable_lora import lora as sl

    class CLIPTextEmbeddings(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.token_embedding = torch.nn.Embedding(50, 16)
            self.position_embedding = torch.nn.Embedding(7, 16)

    class TE(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embeddings = CLIPTextEmbeddings()

    torch.manual_seed(0)
    te = TE()
    w0 = te.embeddings.token_embedding.weight
    sl.add_lora_to(te, target_module=["CLIPTextEmbeddings"], search_class=[torch.nn.Linear, torch.nn.Embedding], r=4)()
    emb = te.embeddings.token_embedding
    assert isinstance(emb, sl.Embedding) and emb.weight is w0 and not emb.weight.requires_grad
    assert emb.lora_A.shape == (4, 50) and emb.lora_B.shape == (16, 4) and emb.lora_A.requires_grad
    assert float(emb.lora_A.abs().max()) == 0 and float(emb.lora_B.abs().max()) > 0
    ids = torch.tensor([[1, 5, 49]])
    assert torch.equal(emb(ids), torch.nn.functional.embedding(ids, w0))          # A = 0: inert at init
    with torch.no_grad():
        emb.lora_A.normal_()
    want = torch.nn.functional.embedding(ids, w0) + (emb.lora_A.t()[ids] @ emb.lora_B.t()) * (4 / 4)
    assert torch.allclose(emb(ids), want, atol=1e-6)
    assert {k for k in sl.lora_state_dict(te)} == {f"embeddings.{t}.lora_{x}" for t in ("token_embedding", "position_embedding") for x in "AB"}


def test_lr_schedules_follow_the_reference_options():
    """train.py:606-612 -> diffusers.get_scheduler: all six names (constant, constant_with_warmup, linear, cosine,
    cosine_with_restarts, polynomial), each against its closed form."""
    import math
    from t2v_amd.training import lr_lambda
    assert [lr_lambda("constant")(k) for k in (0, 10, 10 ** 6)] == [1.0, 1.0, 1.0]
    f = lr_lambda("constant_with_warmup", 4)
    assert [f(k) for k in range(6)] == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0]
    f = lr_lambda("linear", 2, 10)
    assert f(0) == 0.0 and f(1) == 0.5 and f(2) == 1.0 and abs(f(6) - 0.5) < 1e-12 and f(10) == 0.0 and f(12) == 0.0
    f = lr_lambda("cosine", 0, 8)
    assert f(0) == 1.0 and abs(f(4) - 0.5) < 1e-12 and abs(f(8)) < 1e-12
    assert all(f(k) >= f(k + 1) for k in range(8))
    # diffusers' get_cosine_with_hard_restarts_schedule_with_warmup (num_cycles = 1 by default: the cosine half-wave, ending at 0
    # instead of clamping; 2 cycles: back to 1.0 at the half-way restart)
    f = lr_lambda("cosine_with_restarts", 2, 10)
    assert f(1) == 0.5 and f(2) == 1.0 and abs(f(6) - 0.5) < 1e-12 and f(10) == 0.0 and f(11) == 0.0
    f2 = lr_lambda("cosine_with_restarts", 0, 8, num_cycles=2)
    assert f2(0) == 1.0 and abs(f2(2) - 0.5) < 1e-12 and f2(4) == 1.0 and abs(f2(6) - 0.5) < 1e-12 and f2(8) == 0.0
    # diffusers' get_polynomial_decay_schedule_with_warmup (power 1, lr_end 1e-7): lr(k) = (lr0 - lr_end) (1 - (k-w)/(T-w)) + lr_end
    f = lr_lambda("polynomial", 2, 10, base_lr=1e-3)
    assert f(1) == 0.5 and f(2) == 1.0 and abs(f(6) * 1e-3 - ((1e-3 - 1e-7) * 0.5 + 1e-7)) < 1e-15 and abs(f(10) * 1e-3 - 1e-7) < 1e-15
    assert abs(f(50) * 1e-3 - 1e-7) < 1e-15
    import pytest
    with pytest.raises(ValueError):
        lr_lambda("linear", 2)
    with pytest.raises(ValueError):
        lr_lambda("polynomial", 0, 10)                      # needs the optimiser's initial rate
    with pytest.raises(ValueError):
        lr_lambda("one_cycle", 0, 10)


def test_bucket_sizes_of_the_reference_are_whole_latent_octets():
    """`utils/bucketing.py:22-32` (executed from /root/reference by tests/golden/make_golden.py -> buckets.json): starting from a
    configured width / height that is a multiple of 64 pixels, every bucket the reference's datasets resize to is again a multiple
    of 64 pixels = 8 latent cells.  That is why the native UNet may refuse latent grids that are not multiples of 2**num_upsamplers
    (the reference's `upsample_size` branch, models/unet_3d_condition.py:359-367, is never reached from train.py)."""
    import json
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "buckets.json")))
    assert len(cases) >= 4
    for c in cases:
        mw, mh = c["args"][:2]
        if mw % 64 or mh % 64:
            continue
        assert all(int(v) % 64 == 0 for v in c["out"]), c


def test_native_unet_refuses_attention_mask_and_odd_grids():
    """Accepted-and-ignored arguments are refused: `attention_mask` (threaded by the reference, models/unet_3d_condition.py:419-428;
    no bias operand in the native attention core) and latent grids that would need the `upsample_size` path."""
    import inspect
    import t2v_amd.models.unet_3d_condition as m
    src = inspect.getsource(m.UNet3DConditionModel.forward)
    assert "attention_mask is not None" in src and "upsample_size" in src


def test_merge_plan_refreshes_only_the_weights_that_are_read():
    """Default train mode (utils/lora.py:35,89 vs :179): wrappers whose dropout is active never read W_eff, so the merge plan must
    pick the non-dropping entries only, take a projection group as a whole or not at all, and mark every skipped entry stale
    (merge_scale None) so that a layer can never read a merged weight that was not refreshed.  Host logic only (no device)."""
    import torch.nn as nn
    from t2v_amd.lora_bank import LoraEntry, LoraGroup, MergePlan

    class Wrapper(nn.Module):
        def __init__(self, p):
            super().__init__()
            self.dropout = nn.Dropout(p) if p is not None else nn.Identity()
            self.scale = 0.5

    def entry(group=None):
        e = LoraEntry()
        e.group, e.merge_scale = group, None
        return e
    g_mixed, g_off = LoraGroup(), LoraGroup()
    mods = [Wrapper(0.1), Wrapper(0.0), Wrapper(None),            # 0 drops, 1 / 2 do not
            Wrapper(0.0), Wrapper(0.1),                            # 3, 4: one group, one member drops -> neither is merged
            Wrapper(0.0), Wrapper(0.0)]                            # 5, 6: a group that stays merged
    ents = [entry(), entry(), entry(), entry(g_mixed), entry(g_mixed), entry(g_off), entry(g_off)]
    for g, idx in ((g_mixed, (3, 4)), (g_off, (5, 6))):
        g.mods, g.merge_scale = [mods[i] for i in idx], None
    plan = MergePlan.__new__(MergePlan)
    plan.entries = list(zip(ents, mods))
    for m in mods:
        m.train()
    assert plan.wanted() == (1, 2, 5, 6)
    plan._mark(plan.wanted())
    assert [e.merge_scale for e in ents] == [None, 0.5, 0.5, None, None, 0.5, 0.5]
    assert g_mixed.merge_scale is None and g_off.merge_scale == 0.5
    for m in mods:
        m.eval()                                                   # eval_train mode: every Dropout off -> everything merged
    assert plan.wanted() == tuple(range(7))
    plan._mark(plan.wanted())
    assert all(e.merge_scale == 0.5 for e in ents) and g_mixed.merge_scale == 0.5
    plan._mark(())
    assert all(e.merge_scale is None for e in ents) and g_off.merge_scale is None


def test_full_size_fixture_tests_collect_last():
    """a gate tripping in the minutes-long full-size fixture tests must not hide the train / UNet / VAE / CLIP tests from a
    `pytest -x` run — they live in the file that sorts last, and no other GPU test file loads a full-size fixture."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(f for f in os.listdir(here) if f.startswith("test_") and f.endswith(".py"))
    assert files[-1] == "test_zz_fullsize_gpu.py"
    needles = ("_load_" + "fixture(", "fixture_" + "path(")          # (split: this file must not match itself)
    for f in files[:-1]:
        src = open(os.path.join(here, f)).read()
        assert not any(n in src for n in needles), f
