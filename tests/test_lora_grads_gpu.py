"""Model-level parity of the FUSED LoRA path (GPU): eps-MSE and every LoRA factor gradient of the native trainer against the
CPU fp32 oracle (autograd through utils/lora.py:57-62,134-139,211-216), toy config and full-size ModelScope-1.7B shapes.

The native gradients are read through each Parameter's `.grad` (a view of the trainer's flat buffer in the parameter's own
layout), so the lora_bank storage plan — GEMM-layout down factors, transposed / block-diagonal up factors, projection
groups, side-stream factor-gradient launches — is checked against autograd, not against itself.

Tolerances are anchored to the noise floor of the REFERENCE's own recipe: the same oracle under
`torch.autocast(cpu, bfloat16)` vs fp32 (scripts/autocast_floor.py -> tests/golden/autocast_floor_*.json).  The native
path stores activations in bf16 exactly like that recipe, so it is held to `FLOOR_FACTOR` x the recipe's own error (and the
north-star 1e-3 on the loss wherever the recipe itself meets it).
"""
import json
import os

import pytest
import torch

import parity_utils as pu

pytestmark = pytest.mark.gpu
FLOOR_FACTOR = 2.0
RESULTS = os.path.join(os.path.dirname(pu.GOLDEN), "..", "gpurun_out", "parity_r04.jsonl")


def _floor(config, scale):
    with open(os.path.join(pu.GOLDEN, f"autocast_floor_{config}.json")) as f:
        rows = json.load(f)["rows"]
    row = min(rows, key=lambda r: abs(r["lora_up_scale"] - scale))
    return row


def _record(**kw):
    try:
        os.makedirs(os.path.dirname(RESULTS), exist_ok=True)
        with open(RESULTS, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass
    print(json.dumps(kw))


def _set_lora_up(ounet, dunet, scale):
    from oracle.weights import randomize_lora_up
    randomize_lora_up(ounet, scale=scale)
    od = dict(ounet.named_parameters())
    with torch.no_grad():
        for n, p in dunet.named_parameters():
            if p.requires_grad:
                p.copy_(od[n])          # p.data is a view of the trainer's flat fp32 buffer


# ------------------------------------------------------------------------------------------------ toy config, live oracle
@pytest.fixture(scope="module")
def toy():
    from t2v_amd.training import DenoiseTrainer
    ounet, ovae, n = pu.build_oracle(False, 4, 0.05)
    dunet, dvae = pu.build_native(ounet, ovae, False, 4)
    params = [p for p in dunet.parameters() if p.requires_grad]
    return ounet, ovae, dunet, dvae, DenoiseTrainer(dunet, dvae, params, lr=1e-3)


@pytest.mark.parametrize("scale", [0.0, 0.05, 0.2])
def test_toy_lora_factor_gradients_match_oracle(toy, scale):
    from oracle.weights import synthetic_batch
    ounet, ovae, dunet, dvae, trainer = toy
    _set_lora_up(ounet, dunet, scale)
    batch = synthetic_batch(4, 64, 64, seed=100, text_dim=64)
    lo, go = pu.oracle_loss_and_grads(ounet, ovae, batch)
    ld, gd = pu.native_loss_and_grads(trainer, dunet, batch)
    assert set(go) == set(gd)
    c = pu.compare_grads(go, gd)
    fl = _floor("toy", scale)
    _record(test="toy_grads", scale=scale, loss_oracle=lo, loss_native=ld, loss_rel=abs(ld - lo) / abs(lo), floor_loss_rel=fl["loss_rel"],
            floor_grad_rel=fl["grad_rel"], **c)
    assert abs(ld - lo) / abs(lo) < 4e-3                      # toy clip: 64x fewer elements than C1 (see test_train_gpu.py)
    assert c["rel"] < FLOOR_FACTOR * max(fl["grad_rel"], 0.1)
    assert c["cos"] > 0.96
    # per-tensor: the noisiest factors sit at the 2x2-pixel level (16 rows per pass); the recipe itself reads cos 0.95 / rel 0.32
    # there.  A mis-laid-out or missing gradient reads cos ~0 (this assert found the dead text-K/V projection-group node).
    assert c["worst_cos"] > 0.5, "a LoRA factor gradient is decorrelated from autograd's: storage-plan / layout error"
    assert c["tensors"] > 100


def test_toy_optimizer_update_matches_oracle(toy):
    """One full step (backward, global-norm clip, AdamW: train.py:861-879): compare the UPDATE p_after - p_before of every
    LoRA factor, not the parameters (down ~ N(0,1/r) would dominate that norm).  AdamW's first step is
    -lr * g / (|g| + eps): sign-like, so coordinates whose gradient is below the bf16 noise flip; the comparison is therefore
    made on the coordinates that carry the update's signal (|g_oracle| above the per-tensor median) and via the cosine."""
    from oracle.weights import synthetic_batch
    ounet, ovae, dunet, dvae, trainer = toy
    _set_lora_up(ounet, dunet, 0.05)
    oparams = [p for p in ounet.parameters() if p.requires_grad]
    oopt = torch.optim.AdamW(oparams, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    trainer.opt.exp_avg.zero_(); trainer.opt.exp_avg_sq.zero_(); trainer.opt.step_count.zero_()
    before_o = {n: p.detach().clone() for n, p in ounet.named_parameters() if p.requires_grad}
    before_d = {n: p.detach().float().cpu().clone() for n, p in dunet.named_parameters() if p.requires_grad}
    batch = synthetic_batch(4, 64, 64, seed=100, text_dim=64)
    lo, go = pu.oracle_loss_and_grads(ounet, ovae, batch)
    torch.nn.utils.clip_grad_norm_(list(ounet.parameters()), 1.0)
    oopt.step()
    trainer.train_step({k: v.cuda() for k, v in batch.items()})
    torch.cuda.synchronize()
    od = dict(ounet.named_parameters())
    num = den = dot = dd = 0.0
    agree = total = 0
    for n, p in dunet.named_parameters():
        if not p.requires_grad:
            continue
        uo = (od[n].detach() - before_o[n]).double().flatten()
        ud = (p.detach().float().cpu() - before_d[n]).double().flatten()
        g = go[n].double().flatten().abs()
        if float(g.max()) == 0.0:
            continue
        strong = g >= g.median()
        agree += int((torch.sign(uo[strong]) == torch.sign(ud[strong])).sum()); total += int(strong.sum())
        num += float((ud - uo).pow(2).sum()); den += float(uo.pow(2).sum()); dot += float((uo * ud).sum()); dd += float(ud.pow(2).sum())
    rel, cos, sign = (num / den) ** 0.5, dot / (den * dd) ** 0.5, agree / max(total, 1)
    _record(test="toy_update", update_rel=rel, update_cos=cos, sign_agreement_strong=sign)
    assert cos > 0.85 and sign > 0.95, (rel, cos, sign)


# ------------------------------------------------------------------------------------------------ full size, fixtures
def _load_fixture(config, scale, ounet, ovae):
    path = pu.fixture_path(config, scale)
    if not os.path.exists(path):
        return None
    fx = torch.load(path, weights_only=False)
    cs = pu.weight_checksum(ounet, ovae)
    if abs(cs - fx["checksum"]) > 1e-6 * abs(fx["checksum"]):
        print(f"[parity] fixture {os.path.basename(path)}: weight checksum differs ({cs} vs {fx['checksum']}); running the oracle live")
        return None
    return fx


def _compare_with_fixture(fx, gd):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_step", os.path.join(pu.GOLDEN, "make_oracle_step.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    # (1) complete sketch: 4 random +-1 projections of every tensor
    num = den = 0.0
    worst_big, worst_name = 0.0, None
    gn2 = fx["grad_norm"] ** 2
    for n, s_ref in fx["sketches"].items():
        s_dut = mk.sketch(n, gd[n])
        e = float((s_dut.double() - s_ref.double()).pow(2).sum())
        num += e
        den += float(s_ref.double().pow(2).sum())
        nn2 = fx["grad_norms"][n] ** 2
        # per-tensor estimate from 4 projections (E|proj|^2 = ||g||^2): a chi^2_4 variable, +-2x — only tensors holding
        # >= 1 % of the gradient norm are judged one by one (a decorrelated tensor reads ~1.4 on average)
        if nn2 >= 1e-4 * gn2 and (e / (mk.NPROJ * nn2)) ** 0.5 > worst_big:
            worst_big, worst_name = (e / (mk.NPROJ * nn2)) ** 0.5, n
    print(f"[parity] worst sketched tensor: {worst_name} rel~{worst_big:.3f}")
    sk_rel = (num / den) ** 0.5
    # (2) norms of every tensor
    # (round-2 review: the window was 0.6 .. 1.6 — a tensor scaled by 1.5 passed; measured extremes are printed below)
    ratios = {n: float(gd[n].double().norm()) / v for n, v in fx["grad_norms"].items() if v * v >= 1e-6 * gn2}
    print(f"[parity] per-tensor norm ratio native/oracle over {len(ratios)} tensors: min {min(ratios.values()):.3f} "
          f"max {max(ratios.values()):.3f}")
    # window: 0.8 .. 1.25 at the amplitudes a trained LoRA lives at (measured 0.97 .. 1.03); at lora_up ~ 0.2 the recipe's own
    # bf16 run is already 6 % off in the whole gradient (floor 0.06) and single tensors scatter with the rounding path — two
    # tile tables of the same build measured max ratios 1.20 and 1.41 there — so that amplitude keeps round 2's 0.6 .. 1.6
    lo, hi = (0.8, 1.25) if fx.get("lora_up_scale", 0.0) < 0.1 else (0.6, 1.6)
    bad_norm = [(n, r * fx["grad_norms"][n], fx["grad_norms"][n]) for n, r in ratios.items() if not (lo < r < hi)]
    # (3) exact values of the sampled tensors
    go = {n: v for n, v in fx["samples"].items()}
    gs = {n: gd[n].flatten()[: v.numel()] for n, v in fx["samples"].items()}
    c = pu.compare_grads(go, gs, share=0.0)
    return sk_rel, worst_big, bad_norm, c


def _full_case(config, scales):
    from oracle.weights import synthetic_batch
    from t2v_amd.training import DenoiseTrainer
    frames, H, W, r = pu.CONFIGS[config]
    ounet, ovae, n = pu.build_oracle(True, r, scales[0])
    assert n == 574
    dunet, dvae = pu.build_native(ounet, ovae, True, r)
    trainer = DenoiseTrainer(dunet, dvae, [p for p in dunet.parameters() if p.requires_grad], lr=5e-6)
    batch = synthetic_batch(frames, H, W, seed=1234)
    out = []
    for scale in scales:
        _set_lora_up(ounet, dunet, scale)
        fx = _load_fixture(config, scale, ounet, ovae)
        ld, gd = pu.native_loss_and_grads(trainer, dunet, batch)
        if fx is not None:
            lo = fx["loss"]
            sk_rel, worst_big, bad_norm, c = _compare_with_fixture(fx, gd)
        else:
            lo, go = pu.oracle_loss_and_grads(ounet, ovae, batch, single_pass_doubled=(config != "c1"))
            c = pu.compare_grads(go, gd)
            sk_rel, worst_big, bad_norm = c["rel"], c["worst_rel"], []
        fl = _floor("c1", scale)
        row = dict(test=f"full_{config}", scale=scale, loss_oracle=lo, loss_native=ld, loss_rel=abs(ld - lo) / abs(lo),
                   grad_rel_sketch=sk_rel, worst_tensor_rel_sketch=worst_big, norm_outliers=len(bad_norm), sample_rel=c["rel"],
                   sample_cos=c["cos"], sample_worst_cos=c["worst_cos"], floor_loss_rel=fl["loss_rel"], floor_grad_rel=fl["grad_rel"],
                   fixture=fx is not None)
        _record(**row)
        out.append((row, bad_norm))
    return out


# one sketched tensor (>= 1 % of the gradient norm), relative error estimated from its 4 projections: a decorrelated (mis-laid-out)
# tensor reads ~1.4; measured 0.13 - 0.27 over every full-size fixture of rounds 3 and 4 (profiles/r0*_parity.jsonl); the bar was
# 0.5 in round 3 — a tensor 40 % off passed
WORST_TENSOR_BAR = 0.35


def _assert_case(row, bad_norm):
    fl_loss, fl_grad = row["floor_loss_rel"], row["floor_grad_rel"]
    assert row["loss_rel"] < max(1e-3, FLOOR_FACTOR * fl_loss), row       # north-star bar: 1e-3 (where the recipe itself meets it)
    assert row["grad_rel_sketch"] < FLOOR_FACTOR * max(fl_grad, 0.05), row
    assert row["worst_tensor_rel_sketch"] < WORST_TENSOR_BAR, row                       # a decorrelated (mis-laid-out) tensor reads ~1.4;
                                                                           # measured 0.13 - 0.27 (profiles/r03_parity.jsonl)
    assert not bad_norm, bad_norm[:5]
    assert row["sample_cos"] > 0.97 and row["sample_worst_cos"] > 0.5, row


def test_full_c1_loss_and_lora_gradients():
    """ModelScope-1.7B shapes, config C1 (8 frames @128x128, LoRA r=4): LoRA `up` amplitudes 0 (the reference's init,
    utils/lora.py:55), 0.02 and 0.2 of N(0, 1/r); the N(0,1/r) point itself is reported, not asserted: there the network
    leaves its trained regime (loss ~32) and the reference's OWN bf16 recipe has a gradient error of 5.75 (floor file)."""
    rows = _full_case("c1", [0.0, 0.02, 0.2, 1.0])
    for row, bad in rows[:3]:
        _assert_case(row, bad)
    assert rows[3][0]["loss_rel"] < 5e-2


def test_full_c2_loss_and_lora_gradients():
    """The configuration the metric is quoted on (BASELINE.json configs[1]): 16 frames @256x256, LoRA r=16."""
    (row, bad), = _full_case("c2", [0.02])
    _assert_case(row, bad)


# ---- the reference's DEFAULT train mode: LoRA dropout 0.1 (utils/lora.py:35,89) + TemporalConvLayer dropout 0.1
# (models/unet_3d_blocks.py:312; `eval_train` is opt-in, train.py:779-781).  The native masks are counter-based (csrc/common.h);
# oracle/dropout.py restates seed, epoch and element index of every site, so the ORACLE runs the very same masks.
def _dropout_pair(scale, r=4):
    from oracle import dropout as odrop
    from t2v_amd.models import leaves
    from t2v_amd.training import DenoiseTrainer
    import parity_utils as pu
    ounet, ovae, _ = pu.build_oracle(False, r, scale)
    dunet, dvae = pu.build_native(ounet, ovae, False, r)
    for net in (ounet, dunet):                       # back to the constructors' dropout rates (build_* switch them off)
        pu.enable_reference_dropout(net)
    base = pu.DROPOUT_BASE_SEED
    leaves.set_dropout_seed(base)
    params = [p for p in dunet.parameters() if p.requires_grad]
    trainer = DenoiseTrainer(dunet, dvae, params, lr=1e-3)
    # first _fwd_bwd of a fresh trainer: device epoch = (rank << 32) + 2, host step 0
    ctx = odrop.install_protocol(ounet, base, step=0, epoch=2, batch=1, frames=4, passes=2)
    return ounet, ovae, dunet, dvae, trainer, ctx


@pytest.mark.parametrize("scale", [0.05, 0.2])
def test_toy_default_train_mode_with_dropout_matches_oracle(scale):
    """Loss and every LoRA-factor gradient of one train step WITH the reference's default dropout, native (fused masked rank
    update, GroupNorm-epilogue dropout, graph-safe epoch) vs the fp32 oracle running the restated masks."""
    import parity_utils as pu
    from oracle.weights import synthetic_batch
    ounet, ovae, dunet, dvae, trainer, ctx = _dropout_pair(scale)
    batch = synthetic_batch(4, 64, 64, seed=321, text_dim=64)
    l_ref, g_ref = pu.oracle_loss_and_grads(ounet, ovae, batch)
    assert ctx["k"] == 1                                   # both passes ran through the protocol
    l_dut, g_dut = pu.native_loss_and_grads(trainer, dunet, batch)
    rel = abs(l_dut - l_ref) / abs(l_ref)
    cmp = pu.compare_grads(g_ref, g_dut)
    big = pu.compare_grads(g_ref, g_dut, share=1e-2)        # tensors holding >= 1 % of the gradient norm, judged one by one
    print(f"dropout mode, lora_up~{scale}: loss oracle {l_ref:.6f} native {l_dut:.6f} rel {rel:.2e}; grads rel {cmp['rel']:.3f} "
          f"cos {cmp['cos']:.4f} worst tensor rel {cmp['worst_rel']:.3f} cos {cmp['worst_cos']:.3f} over {cmp['tensors']}; "
          f"among the {big['tensors']} tensors >= 1 % of the norm: worst cos {big['worst_cos']:.3f}")
    # same gates as the dropout-free toy comparison above (toy clip: 64x fewer latent elements than C1).  Tensors below 1 % of
    # the gradient norm are sums of 8-32 bf16 products at the 1x1 / 2x2 levels of the toy grid: with every wrapper dropping
    # (Conv3d wrappers included) one of them can decorrelate (measured 0.55 at 0.3 % of the norm) — they are gated by the
    # dropout-free comparison's 0.5 bar, the ones that matter by 0.8 (measured 0.87 - 0.99); the full-size gate is the C1 fixture test below.
    assert rel < 4e-3
    assert cmp["rel"] < 0.25 and cmp["cos"] > 0.97
    assert big["worst_cos"] > 0.8 and cmp["worst_cos"] > 0.5
    # and the masks matter: with the protocol switched off in the oracle the losses must differ visibly
    for m in ounet.modules():
        if m.__class__.__name__ == "ProtocolDropout":
            m.p = 0.0
    l_off, g_off = pu.oracle_loss_and_grads(ounet, ovae, batch)
    off = pu.compare_grads(g_ref, g_off)
    print(f"oracle with its masks switched off: loss moves {abs(l_off - l_ref) / abs(l_ref):.2e}, gradients rel {off['rel']:.3f}")
    # (the size of the loss shift depends on the drawn masks: 2e-3 .. 7e-3 over the two protocol versions; the gradients of the
    #  dropped branches move by ~sqrt(p / (1 - p)) whatever the draw)
    assert abs(l_off - l_ref) / abs(l_ref) > 3 * max(rel, 1e-4)
    assert off["rel"] > cmp["rel"]


def test_toy_rank_beyond_the_merge_window_trains_and_matches_oracle():
    """lora_rank 40 (padded rank 40 > the merge kernel's 32-wide window): the trainer must still construct — those layers keep
    their LoRA branch apart (functional.lora_layer) — and loss / factor gradients must match the oracle like the merged path."""
    import parity_utils as pu
    from oracle.weights import synthetic_batch
    from t2v_amd.training import DenoiseTrainer
    ounet, ovae, _ = pu.build_oracle(False, 40, 0.05)
    dunet, dvae = pu.build_native(ounet, ovae, False, 40)
    params = [p for p in dunet.parameters() if p.requires_grad]
    trainer = DenoiseTrainer(dunet, dvae, params, lr=1e-3)
    big = [e for e, _ in [(m._t2v_bank, m) for m in dunet.modules() if getattr(m, "_t2v_bank", None) is not None] if e.rp > 32]
    assert big and all(getattr(e, "merge_scale", None) is None for e in big)
    batch = synthetic_batch(4, 64, 64, seed=100, text_dim=64)
    lo, go = pu.oracle_loss_and_grads(ounet, ovae, batch)
    ld, gd = pu.native_loss_and_grads(trainer, dunet, batch)
    c = pu.compare_grads(go, gd)
    print(f"r=40: loss oracle {lo:.6f} native {ld:.6f}; grads rel {c['rel']:.3f} cos {c['cos']:.4f} worst cos {c['worst_cos']:.3f}")
    assert abs(ld - lo) / abs(lo) < 4e-3
    assert c["rel"] < 0.25 and c["cos"] > 0.96 and c["worst_cos"] > 0.5
    loss = trainer.train_step({k: v.cuda() for k, v in batch.items()})       # and a whole step runs
    assert torch.isfinite(loss)


@pytest.mark.parametrize("config", ["c3", "c3full"])
def test_c3_full_finetune_gradients_match_the_oracle_fixture(config):
    """Config C3 (BASELINE.json configs[2], train.py:172-236: every UNet parameter trainable, no LoRA) at FULL model size: loss
    and the gradient of all 1.41 B parameters — complete +-1 sketch, per-tensor norms, exact samples — against the committed
    CPU-oracle fixtures (tests/golden/make_oracle_step.py --config c3: the C1 clip; --config c3full: configs[2]'s own clip,
    16 frames @256x256).  The weight gradients come from the K-major GEMM family (dW = x^T dy), which no LoRA configuration
    exercises at full size."""
    from oracle.weights import synthetic_batch
    from t2v_amd.training import DenoiseTrainer
    frames, H, W, _ = pu.CONFIGS[config]
    ounet, ovae = pu.build_oracle_full_finetune(True)
    fx = _load_fixture(config, 0.0, ounet, ovae)
    assert fx is not None, f"tests/golden/oracle_step_{config}_s0.pt is missing or was made for other weights"
    dunet, dvae = pu.build_native_full_finetune(ounet, ovae)
    del ounet, ovae
    trainer = DenoiseTrainer(dunet, dvae, list(dunet.parameters()), lr=5e-6)
    assert trainer.opt.merge is None and trainer.opt.numel > 1.4e9
    batch = synthetic_batch(frames, H, W, seed=1234)
    ld, gd = pu.native_loss_and_grads(trainer, dunet, batch)
    sk_rel, worst_big, bad_norm, c = _compare_with_fixture(fx, gd)
    row = dict(test=f"full_{config}", scale=0.0, loss_oracle=fx["loss"], loss_native=ld, loss_rel=abs(ld - fx["loss"]) / abs(fx["loss"]),
               grad_rel_sketch=sk_rel, worst_tensor_rel_sketch=worst_big, norm_outliers=len(bad_norm), sample_rel=c["rel"],
               sample_cos=c["cos"], sample_worst_cos=c["worst_cos"], tensors=len(fx["sketches"]), fixture=True)
    _record(**row)
    print(row)
    assert row["loss_rel"] < 1e-3, row
    assert row["grad_rel_sketch"] < 0.15 and row["worst_tensor_rel_sketch"] < WORST_TENSOR_BAR, row
    assert not bad_norm, bad_norm[:5]
    assert row["sample_cos"] > 0.97 and row["sample_worst_cos"] > 0.5, row


@pytest.mark.parametrize("config", ["c1", "c2"])
def test_default_train_mode_with_dropout_matches_the_oracle_fixture(config):
    """The reference's DEFAULT train mode at FULL model size — config C1 and the benchmark configuration C2 (16 frames @256x256,
    r = 16: what `python bench.py` times): LoRA dropout 0.1 on the Linear / Conv2d wrappers + TemporalConvLayer dropout 0.1, two
    passes with their own masks.  The fixtures are the CPU fp32 oracle running the restated masks of the native protocol
    (tests/golden/make_oracle_step.py --config c1|c2 --scales 0.02 --dropout); compared: loss, the complete sketch of every factor
    gradient, per-tensor norms, sampled tensors.  This is the path with the LoRA branch folded into the base launches' epilogues
    (T2VGemm.lr_mode), the masked dt / dU kernels and the GroupNorm-epilogue dropout."""
    from oracle.weights import synthetic_batch
    from t2v_amd.models import leaves
    from t2v_amd.training import DenoiseTrainer
    frames, H, W, r = pu.CONFIGS[config]
    scale = 0.02
    ounet, ovae, n = pu.build_oracle(True, r, scale)
    path = pu.fixture_path(config, scale, dropout=True)
    assert os.path.exists(path), path
    fx = torch.load(path, weights_only=False)
    assert fx.get("dropout") and abs(pu.weight_checksum(ounet, ovae) - fx["checksum"]) <= 1e-6 * abs(fx["checksum"])
    dunet, dvae = pu.build_native(ounet, ovae, True, r)
    del ounet, ovae
    pu.enable_reference_dropout(dunet)
    leaves.set_dropout_seed(pu.DROPOUT_BASE_SEED)
    trainer = DenoiseTrainer(dunet, dvae, [p for p in dunet.parameters() if p.requires_grad], lr=5e-6)
    assert trainer.opt.prep is not None and trainer.opt.prep.wanted()
    batch = synthetic_batch(frames, H, W, seed=1234)
    ld, gd = pu.native_loss_and_grads(trainer, dunet, batch)      # first step of a fresh trainer: epoch 2, host step 0
    sk_rel, worst_big, bad_norm, c = _compare_with_fixture(fx, gd)
    row = dict(test=f"full_{config}_dropout", scale=scale, loss_oracle=fx["loss"], loss_native=ld, loss_rel=abs(ld - fx["loss"]) / abs(fx["loss"]),
               grad_rel_sketch=sk_rel, worst_tensor_rel_sketch=worst_big, norm_outliers=len(bad_norm), sample_rel=c["rel"],
               sample_cos=c["cos"], sample_worst_cos=c["worst_cos"], fixture=True)
    _record(**row)
    print(row)
    assert row["loss_rel"] < 1e-3, row                     # north_star's bar (a wrong or missing mask moves the loss by several 1e-2)
    assert row["grad_rel_sketch"] < 0.15 and row["worst_tensor_rel_sketch"] < WORST_TENSOR_BAR, row
    assert not bad_norm, bad_norm[:5]
    assert row["sample_cos"] > 0.97 and row["sample_worst_cos"] > 0.5, row
