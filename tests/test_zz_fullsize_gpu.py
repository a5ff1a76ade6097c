"""Full-size ModelScope-1.7B parity against the committed CPU-oracle fixtures (GPU; minutes per test — this file collects LAST so
that no gate here can hide the train / UNet / VAE / CLIP tests from a `pytest -x` run).

eps-MSE and every trainable-tensor gradient of one native `_fwd_bwd` against tests/golden/oracle_step_*.pt
(tests/golden/make_oracle_step.py: the fp32 oracle's two-pass step of train.py:793-834 on host-seeded weights / inputs):
  * configs C1 (amplitudes 0 / 0.02 / 0.2 / N(0,1/r)), C2 (the benchmark configuration), C3 (full finetune, both clips);
  * the reference's DEFAULT train mode (LoRA + TemporalConvLayer dropout, masks restated in the oracle) at C1 and C2.
Compared: the loss, a complete +-1 sketch of every gradient tensor, per-tensor norms, sampled tensors element by element and — the
per-tensor gate — the COMPLETE gradient (or a 64-bucket count-sketch of it, for tensors above 65 536 elements) of every tensor holding
>= 1 % of the gradient norm, each against the error the reference's own bf16-autocast recipe makes on that same tensor where the
fixture records it (`big_floor`).

Why the per-tensor gate changed in round 5: rounds 3-4 estimated ONE tensor's error from its 4 sketch projections — a chi^2_4
variable, 0.4x .. 1.6x the true value — and gated the maximum over ~200 tensors at 0.35.  `down_blocks.2.temp_attentions.0...
attn1.to_q.lora_up` at amplitude 0.2 read 0.389 / 0.395 there (profiles/r03_parity.jsonl, GPUTEST_r04): the fp32-vs-bf16-autocast
ORACLE is 0.294 off on exactly that tensor (the noisiest site of the recipe itself: 16 rows x 8 frames at the 1280-channel level feed
a softmax over 8 keys), measured element by element (fixture `big_floor`).  The gate now uses true errors and that per-tensor yardstick.
"""
import os

import pytest
import torch

import parity_utils as pu
from parity_utils import FLOOR_FACTOR, floor_row as _floor, record as _record, set_lora_up as _set_lora_up

pytestmark = pytest.mark.gpu


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


def _fixture_tools():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_step", os.path.join(pu.GOLDEN, "make_oracle_step.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk


# Per-tensor bars (tensors holding >= 1 % of the gradient norm):
#   fixtures with `big` (round 5): the TRUE relative error ||g_native - g_oracle|| / ||g_oracle|| of the tensor, element by
#   element (64-bucket count-sketch above 65 536 elements: +-9 %), must stay below FLOOR_FACTOR x the error of the reference's own bf16
#   recipe ON THAT TENSOR (`big_floor`, never taken below MIN_TENSOR_FLOOR), or below TRUE_TENSOR_BAR where the fixture has no
#   floor arm (dropout / C2 / C3 fixtures: one oracle run only).  Every committed fixture carries these references
#   (tests/test_parity_floor.py::test_full_size_fixtures_carry_true_per_tensor_references); the 4-projection estimate of rounds
#   3-4 is still printed beside them.
# Measured (profiles/r05_parity.jsonl, identical in the three full runs of round 5): C1 amplitudes 0 / 0.02: worst true error 0.138 /
# 0.137, every tensor BELOW its own bf16-recipe floor (0.13 - 0.17); amplitude 0.2: 0.273 on the to_q tensor above (floor 0.294), worst
# ratio to the floor 1.36 (down_blocks.2.attentions.0 ... to_v: 0.235 vs 0.173); fixtures without a floor arm: C1 default mode 0.127,
# C2 0.146, C2 default mode 0.128, C3 0.153 / 0.152.
FULL_SUITE = os.environ.get("T2V_TEST_FULL", "0") == "1"     # every amplitude / fixture (the round-5 suite: 685 s on one MI355X)
MIN_TENSOR_FLOOR = 0.10
TRUE_TENSOR_BAR = 0.20      # round 6: 0.25 -> 0.20 (measured 0.127 - 0.153 on the five fixtures without a floor arm)


def _compare_with_fixture(fx, gd):
    mk = _fixture_tools()
    # (1) complete sketch: 4 random +-1 projections of every tensor
    num = den = 0.0
    worst_sk, worst_sk_name = 0.0, None
    gn2 = fx["grad_norm"] ** 2
    for n, s_ref in fx["sketches"].items():
        s_dut = mk.sketch(n, gd[n])
        e = float((s_dut.double() - s_ref.double()).pow(2).sum())
        num += e
        den += float(s_ref.double().pow(2).sum())
        nn2 = fx["grad_norms"][n] ** 2
        if nn2 >= 1e-4 * gn2 and (e / (mk.NPROJ * nn2)) ** 0.5 > worst_sk:
            worst_sk, worst_sk_name = (e / (mk.NPROJ * nn2)) ** 0.5, n
    sk_rel = (num / den) ** 0.5
    # (1b) every tensor holding >= 1 % of the gradient norm, judged one by one on its TRUE error
    per = []                                            # (true rel, floor or None, name)
    assert "big" in fx, "fixture without complete per-tensor references: regenerate it (tests/golden/make_oracle_step.py)"
    for n, ref in fx["big"].items():
        a, b = ref.double().flatten(), gd[n].double().flatten()
        per.append((float((b - a).norm() / a.norm()), fx["big_floor"].get(n), n))
    for n, s_ref in fx.get("big_sketch", {}).items():
        s_dut = mk.sketch_big(n, gd[n])
        e = float((s_dut - s_ref.double()).pow(2).sum())
        per.append(((e / fx["grad_norms"][n] ** 2) ** 0.5, fx["big_floor"].get(n), n))
    per.sort(reverse=True)
    print(f"[parity] {len(per)} tensors >= 1 % of the gradient norm, true error (bf16-recipe floor of the same tensor):")
    for r, f, n in per[:6]:
        print(f"[parity]    {r:.3f} ({'-' if f is None else f'{f:.3f}'})  {n}")
    worst_big = per[0][0]
    over = [(r, f, n) for r, f, n in per
            if r >= (TRUE_TENSOR_BAR if f is None else FLOOR_FACTOR * max(f, MIN_TENSOR_FLOOR))]
    ratio = max((r / max(f, MIN_TENSOR_FLOOR) for r, f, n in per if f is not None), default=None)
    print(f"[parity] worst 4-projection estimate: {worst_sk_name} ~{worst_sk:.3f}")
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
    c["tensor_over_bar"] = over
    c["worst_tensor_over_floor"] = ratio
    c["worst_sketch4"] = worst_sk
    c["true_errors"] = True
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
            c["tensor_over_bar"] = [(worst_big, None, "live oracle")] if worst_big >= TRUE_TENSOR_BAR else []
        fl = _floor("c1", scale)
        row = dict(test=f"full_{config}", scale=scale, loss_oracle=lo, loss_native=ld, loss_rel=abs(ld - lo) / abs(lo),
                   grad_rel_sketch=sk_rel, worst_tensor_rel=worst_big, worst_tensor_is_true_error=c.get("true_errors", True),
                   worst_tensor_over_floor=c.get("worst_tensor_over_floor"), worst_sketch4=c.get("worst_sketch4"),
                   tensors_over_bar=[(round(r, 3), n) for r, _, n in c.get("tensor_over_bar", [])], norm_outliers=len(bad_norm), sample_rel=c["rel"],
                   sample_cos=c["cos"], sample_worst_cos=c["worst_cos"], floor_loss_rel=fl["loss_rel"], floor_grad_rel=fl["grad_rel"],
                   fixture=fx is not None)
        _record(**row)
        out.append((row, bad_norm))
    return out


def _assert_case(row, bad_norm):
    fl_loss, fl_grad = row["floor_loss_rel"], row["floor_grad_rel"]
    assert row["loss_rel"] < max(1e-3, FLOOR_FACTOR * fl_loss), row       # north-star bar: 1e-3 (where the recipe itself meets it)
    assert row["grad_rel_sketch"] < FLOOR_FACTOR * max(fl_grad, 0.05), row
    assert not row["tensors_over_bar"], row                               # per-tensor bars: see TRUE_TENSOR_BAR above
    assert not bad_norm, bad_norm[:5]
    assert row["sample_cos"] > 0.97 and row["sample_worst_cos"] > 0.5, row


def test_full_c1_loss_and_lora_gradients():
    """ModelScope-1.7B shapes, config C1 (8 frames @128x128, LoRA r=4): LoRA `up` amplitudes 0 (the reference's init,
    utils/lora.py:55), 0.02 and 0.2 of N(0, 1/r); the N(0,1/r) point itself is reported, not asserted: there the network
    leaves its trained regime (loss ~32) and the reference's OWN bf16 recipe has a gradient error of 5.75 (floor file)."""
    # (round 6: the amplitudes 0.02 and 1.0 run only with T2V_TEST_FULL=1 — the suite must fit the driver's time limit with the
    #  new default-mode grid tests; 0 is the reference's init, 0.2 the hardest asserted point, 0.02 is covered at C2 and in both
    #  default-mode fixtures)
    scales = [0.0, 0.02, 0.2, 1.0] if FULL_SUITE else [0.0, 0.2]
    rows = _full_case("c1", scales)
    for (row, bad), sc in zip(rows, scales):
        if sc < 1.0:
            _assert_case(row, bad)
        else:
            assert row["loss_rel"] < 5e-2


def test_full_c2_loss_and_lora_gradients():
    """The configuration the metric is quoted on (BASELINE.json configs[1]): 16 frames @256x256, LoRA r=16."""
    (row, bad), = _full_case("c2", [0.02])
    _assert_case(row, bad)


@pytest.mark.parametrize("config", [pytest.param("c3", marks=pytest.mark.skipif(not FULL_SUITE, reason="C3 on the C1 clip: T2V_TEST_FULL=1 "
                                                                                "(c3full, C3's own clip, always runs)")), "c3full"])
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
               grad_rel_sketch=sk_rel, worst_tensor_rel=worst_big, worst_tensor_is_true_error=c["true_errors"], worst_sketch4=c["worst_sketch4"],
               tensors_over_bar=[(round(r, 3), n) for r, _, n in c["tensor_over_bar"]], norm_outliers=len(bad_norm), sample_rel=c["rel"],
               sample_cos=c["cos"], sample_worst_cos=c["worst_cos"], tensors=len(fx["sketches"]), fixture=True)
    _record(**row)
    print(row)
    assert row["loss_rel"] < 1e-3, row
    assert row["grad_rel_sketch"] < 0.15 and not row["tensors_over_bar"], row
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
               grad_rel_sketch=sk_rel, worst_tensor_rel=worst_big, worst_tensor_is_true_error=c["true_errors"], worst_sketch4=c["worst_sketch4"],
               tensors_over_bar=[(round(r, 3), n) for r, _, n in c["tensor_over_bar"]], norm_outliers=len(bad_norm), sample_rel=c["rel"],
               sample_cos=c["cos"], sample_worst_cos=c["worst_cos"], fixture=True)
    _record(**row)
    print(row)
    assert row["loss_rel"] < 1e-3, row                     # north_star's bar (a wrong or missing mask moves the loss by several 1e-2)
    assert row["grad_rel_sketch"] < 0.15 and not row["tensors_over_bar"], row
    assert not bad_norm, bad_norm[:5]
    assert row["sample_cos"] > 0.97 and row["sample_worst_cos"] > 0.5, row
