#!/usr/bin/env python
"""bench.py — train-step videos/sec of the MI355X-native denoising path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one optimisation step of the reference's train loop on one 16-frame 256x256 clip per GPU
(config C2: ModelScope-1.7B shapes, LoRA rank 16 on the whole UNet, bf16 compute, fp32 LoRA/optimizer state):
VAE encode of the (1,16,3,256,256) frame stack -> noise + add_noise -> 2x UNet forward -> eps-MSE -> backward ->
RCCL all-reduce of the flat LoRA gradient -> global-norm clip -> fused AdamW (train.py:720-836,848-879).
Synthetic data and random-init weights of the ModelScope architecture (no dataset/checkpoint exists offline).
Rank 0 prints ONE JSON line (contract in the task statement) including `roofline` for the dominant kernel family
(the MFMA GEMM core: every Linear/Conv2d/Conv3d forward, backward-data and weight-gradient launch of a step,
timed with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIGS = {
    # name: (frames, height, width, lora_rank)   rank 0 = full UNet finetune (no LoRA)
    "c2": (16, 256, 256, 16),     # BASELINE.json configs[1] — the configuration the metric is quoted on
    "c1": (8, 128, 128, 4),       # configs[0] — the reference's CPU-runnable case
    "c3": (16, 256, 256, 0),      # configs[2] — full UNet finetune (1.41 B trainable), gradient checkpointing off
    "c4": (24, 320, 576, 16),     # configs[3] — one clip of the Zeroscope-576w shape (24 frames @576x320, latent 40x72) per GPU
    "c5": (24, 576, 1024, 32),    # configs[4] — one clip of the Zeroscope-XL shape (24 frames @1024x576, latent 72x128), r=32,
}                                 #              CACHED latents (the VAE is not in the step); run it with --grad-checkpointing
CACHED_LATENTS = {"c5"}
METRICS = {
    "c2": "train-step videos/sec (16-frame 256x256, ModelScope-1.7B LoRA)",
    "c1": "train-step videos/sec (8-frame 128x128, ModelScope-1.7B LoRA r=4)",
    "c3": "train-step videos/sec (16-frame 256x256, ModelScope-1.7B full UNet finetune)",
    "c4": "train-step videos/sec (24-frame 576x320, Zeroscope-576w shape, LoRA r=16)",
    "c5": "train-step videos/sec (24-frame 1024x576, Zeroscope-XL shape, LoRA r=32, cached latents)",
}
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E peak, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps (median is reported; one untimed warm-up first)")
    ap.add_argument("--no-cpu-c2", action="store_true", help="skip the single timed CPU-oracle step at the C2 clip (SURVEY 8(d): C1 x3 and C2 once)")
    ap.add_argument("--eval-train", action="store_true",
                    help="the reference's OPT-IN eval_train mode (train.py:779-781: every Dropout off) instead of its default train "
                         "mode, which is what the headline measures: LoRA dropout 0.1 (utils/lora.py:35,89; lora_unet_dropout 0.1 in "
                         "every shipped YAML) and TemporalConvLayer dropout 0.1 (models/unet_3d_blocks.py:312)")
    ap.add_argument("--dropout", action="store_true", help="(default since round 4; kept so that older command lines still parse)")
    ap.add_argument("--grad-checkpointing", action="store_true", help="train.py:127-129,670-675")
    ap.add_argument("--no-text-encoder", action="store_true", help="feed synthetic text states instead of running CLIP")
    ap.add_argument("--no-default-mode", "--no-other-mode", dest="no_default_mode", action="store_true",
                    help="skip the extra short run in the OTHER dropout mode (eval_train beside the default-mode headline, or the "
                         "default mode beside an --eval-train line) that fills config.eval_train_ms_per_step / default_mode_ms_per_step")
    ap.add_argument("--no-host-timing", action="store_true",
                    help="skip the five extra steps that measure config.host_ms_per_step (profiling scripts: the trace then ends with "
                         "exactly the timed replays)")
    ap.add_argument("--export-tune-table", default=None,
                    help="write the GEMM tile table after the run (use with T2V_GEMM_AUTOTUNE=live; scripts/tune_gemm_table.sh)")
    return ap.parse_args()


def build_models(frames, lora_rank, device, seed, dropout=False, grad_ckpt=False):
    import t2v_amd  # noqa: F401
    from t2v_amd.models.unet_3d_condition import UNet3DConditionModel
    from t2v_amd.models.vae import AutoencoderKL
    from t2v_amd.utils.lora_handler import LoraHandler
    torch.manual_seed(seed)
    with torch.device(device):
        unet = UNet3DConditionModel()
        vae = AutoencoderKL()
        for m in unet.modules():                        # upstream zero-inits conv4; draw it so the Conv3d path is live
            if m.__class__.__name__ == "TemporalConvLayer":
                c = m.conv4[-1].weight.shape[1]
                torch.nn.init.normal_(m.conv4[-1].weight, std=(3 * c) ** -0.5)
    vae.requires_grad_(False)
    if lora_rank > 0:
        unet.requires_grad_(False)
        handler = LoraHandler(use_unet_lora=True)
        handler.add_lora_to_model(True, unet, ["UNet3DConditionModel"], 0.1 if dropout else 0.0, None, r=lora_rank)
    else:
        unet.requires_grad_(True)                       # config C3: every UNet parameter trains (train.py:172-236 param groups)
    unet.train()
    if not dropout:
        for m in unet.modules():                        # the reference's `eval_train` mode (train.py:779-781): dropout off
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    if grad_ckpt:
        unet.enable_gradient_checkpointing()
    vae.eval()
    trainable = [p for p in unet.parameters() if p.requires_grad]
    return unet, vae, trainable


def build_text_encoder(device):
    """Random-init CLIP text tower in the ModelScope/OpenCLIP ViT-H shape (23 layers, d=1024, 16 heads, MLP 4096, 77 tokens;
    SURVEY.md A.10), frozen, bf16.  It is adjacent to the hot path (§8f row 2, ~45 GFLOP of a 24 TFLOP step); since round 4 its
    forward runs on this library's kernels (t2v_amd/models/clip_text.py: LayerNorm, fused q/k/v on the skinny weight-streaming
    kernel, causal attention, GELU — only the two embedding gathers are torch index ops); it is inside the timed step so that no
    part of the reference's step is skipped."""
    try:
        from transformers import CLIPTextConfig, CLIPTextModel
        cfg = CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                             num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu", projection_dim=1024)
        torch.manual_seed(2)
        with torch.device(device):
            te = CLIPTextModel(cfg)
        return te.to(torch.bfloat16).eval().requires_grad_(False)
    except Exception as e:   # noqa: BLE001
        print(f"[bench] CLIP text encoder unavailable ({type(e).__name__}: {e}); using synthetic text states", file=sys.stderr)
        return None


def synthetic_batch(frames, H, W, device, seed, with_ids=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    b = dict(pixel_values=(torch.rand(1, frames, 3, H, W, generator=g) * 2 - 1).to(device))
    if with_ids:   # utils/dataset.py:43-52: [BOS] + k tokens + [EOS] padding to 77, with the dataset's extra leading dim
        ids = torch.full((1, 1, 77), 49407, dtype=torch.long)
        ids[0, 0, 0] = 49406
        ids[0, 0, 1:9] = torch.randint(0, 49406, (8,), generator=g)
        b["prompt_ids"] = ids.to(device)
    else:
        b["encoder_hidden_states"] = torch.randn(1, 77, 1024, generator=g).to(device)
    return b


class _KernelEvents:
    """Raw hipEvent_t pairs for the library's measurement hook (t2v_gemm_timing_events): the runtime writes the kernel's own
    begin / end timestamps into them (hipExtLaunchKernelGGL), i.e. the duration a rocprofv3 kernel trace reports."""

    def __init__(self):
        import ctypes
        self.C = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.made = []

    def pair(self):
        out = []
        for _ in range(2):
            ev = self.C.c_void_p()
            if self.hip.hipEventCreate(self.C.byref(ev)) != 0:
                return None
            self.made.append(ev)
            out.append(ev)
        return tuple(out)

    def ms(self, pair):
        v = self.C.c_float()
        return float(v.value) if self.hip.hipEventElapsedTime(self.C.byref(v), pair[0], pair[1]) == 0 else None

    def close(self):
        for ev in self.made:
            self.hip.hipEventDestroy(ev)
        self.made = []


def gemm_roofline(trainer, batch):
    """One instrumented eager step.  Every GEMM-family launch is timed twice on its launch stream: by the kernel's own begin /
    end timestamps (hipExtLaunchKernelGGL through the library's measurement hook — what a kernel trace reports; single-kernel
    NN launches) and by an event pair recorded around the call (which also contains the dispatch gap, ~5 us per launch in an
    eager pass; used where the hook does not apply: split-K pairs, K-major kernels)."""
    import t2v_amd.functional as F
    records = []
    orig = F.launch_gemm
    try:
        kev = _KernelEvents()
        lib = F.nv.lib()
    except Exception:   # noqa: BLE001
        kev = None

    def run_timed(call):
        """-> (result, outer start, outer end, kernel-timestamp pair or None)"""
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        inner = kev.pair() if kev is not None else None
        s.record()
        if inner is not None:
            lib.t2v_launch_timing_events(inner[0], inner[1])
        r = call()
        if inner is not None and not lib.t2v_launch_timing_consumed():
            inner = None
        e.record()
        return r, s, e, inner

    def dur(s, e, inner):        # kernel timestamps where the hook applied, the outer event pair otherwise
        v = kev.ms(inner) if (kev is not None and inner is not None) else None
        return v if v is not None and v > 0 else s.elapsed_time(e)

    def timed(cs=None, **kw):
        ret, s, e, inner = run_timed(lambda: orig(cs=cs, **kw))
        z = max(1, kw.get("batch", 1))
        flops = 2.0 * kw["M"] * kw["N"] * kw["K"] * z
        lr_ = kw.get("lr")
        if lr_ is not None:      # rank-wide epilogue term of the launch (dropped LoRA branch): t = x D^T once + the rank-wide product
            flops += 2.0 * kw["M"] * lr_["rp"] * (kw["N"] * lr_.get("taps", 1) + (kw["K"] if lr_["mode"] == 2 else 0))
        geom = kw.get("geom")
        if geom is not None and geom.tdiv == 2:
            flops /= 4.0                                 # 3/4 of the gathered taps are structural zeros
        nn_kernel = not kw.get("a_trans", 0) and not kw.get("b_trans", 0)
        # algorithmic bytes: un-replicated input once + output once + weights once
        taps_ = (geom.KH * geom.KW) if geom is not None else 1
        abytes = (kw["M"] * (kw["K"] // taps_) + kw["M"] * kw["N"] + kw["N"] * kw["K"]) * 2.0 * z
        records.append((flops, s, e, nn_kernel, abytes, inner))
        if tflag["attn"] > 0:                   # (module forward only: the flag is down while autograd runs the backward)
            tunit["gemm"].append((s, e, inner))
        rc = (kw["N"] - kw["n_split"]) if kw.get("n_split", 0) > 0 else 0
        gtag = "lin" if geom is None else f"conv{geom.KH}x{geom.KW}s{geom.sy}d{geom.tdiv}u{geom.up}"
        shapes.append(((kw["M"], kw["N"] - rc, rc, kw["K"], z, gtag, "nn" if nn_kernel else "tn",
                        "res" if kw.get("R") else "-"), flops, s, e, inner))
        if nn_kernel and geom is not None and geom.KH == 3 and geom.KW == 1:
            # the (3,1,1) Conv3d launches (forward + backward-data): SURVEY 8(d) bytes = x once + y once + weights once
            taps = 3
            conv3d.append((flops, (kw["M"] * (kw["K"] // taps) + kw["M"] * kw["N"] + kw["N"] * kw["K"]) * 2.0, s, e, inner, kw["M"]))
        return ret

    orig_pair = F.launch_gemm_pair

    def timed_pair(kw_a, kw_b):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig_pair(kw_a, kw_b)
        e.record()
        records.append((2.0 * (kw_a["M"] * kw_a["N"] * kw_a["K"] + kw_b["M"] * kw_b["N"] * kw_b["K"]), s, e, False, 0.0, None))

    conv3d, attn, wgrad, shapes, norms = [], [], [], [], []
    # SURVEY 8(d)'s fused temporal-attention unit LN -> QKV -> softmax(FxF) -> PV -> out-proj (+residual): its launches are
    # tagged while a TransformerTemporalModel's Attention module is running its FORWARD
    import t2v_amd.models.leaves as _leaves
    tflag = {"temporal": 0, "attn": 0}
    tunit = {"flops": 0.0, "gemm": [], "core": [], "ln": [], "calls": 0}
    _tt_fwd, _at_fwd = _leaves.TransformerTemporalModel.forward, _leaves.Attention.forward

    def tt_forward(self, x, *a, **k):
        tflag["temporal"] += 1
        tflag["frames"] = k.get("num_frames", 1)
        try:
            return _tt_fwd(self, x, *a, **k)
        finally:
            tflag["temporal"] -= 1

    def at_forward(self, x, *a, **k):
        if tflag["temporal"] and torch.is_grad_enabled():
            T, Cc = x.shape[0], self.heads * self.dim_head
            tunit["flops"] += 8.0 * Cc * x.shape[1] * T + 4.0 * T * tflag.get("frames", 1) * Cc
            tunit["calls"] += 1
        tflag["attn"] += 1 if tflag["temporal"] else 0
        try:
            return _at_fwd(self, x, *a, **k)
        finally:
            tflag["attn"] -= 1 if tflag["temporal"] else 0
    wgrad_layers = [0]
    nv = F.nv
    orig_call = nv.call

    NORM = {"t2v_gn_stats": (2, "gn_fwd"), "t2v_gn_apply": (4, "gn_fwd"), "t2v_gn_finish": (None, "gn_finish"), "t2v_gn_bwd_stats": (4, "gn_bwd"),
            "t2v_gn_bwd_apply": (6, "gn_bwd"), "t2v_layernorm_fwd": (None, "ln_fwd"), "t2v_layernorm_bwd": (None, "ln_bwd")}

    def timed_call(name, *a):
        if name not in ("t2v_attn_fwd", "t2v_attn_bwd", "t2v_lora_wgrad", "t2v_lora_wgrad_batch") and name not in NORM:
            return orig_call(name, *a)
        r, s, e, inner = run_timed(lambda: orig_call(name, *a))
        if name in NORM:
            pos, kind = NORM[name]
            if name == "t2v_gn_finish":          # statistics from the GEMM epilogues' column sums: no pass over the activations
                norms.append((kind, 0.0, 0.0, s, e, inner))
                return r
            if pos is None:                      # layernorm_fwd(x,ldx,y,ldy,rows,C,..) / bwd(x,ldx,dy,lddy,dx,lddx,rows,C,..)
                E = float(a[4] * a[5]) if name.endswith("fwd") else float(a[6] * a[7])
                moved = 2 * E * 2 if name.endswith("fwd") else (3 + (1 if a[-3] else 0)) * E * 2
                alg = moved
            else:
                E = float(a[pos]) * a[pos + 1] * a[pos + 2]
                addend = name == "t2v_gn_bwd_apply" and a[-3]
                moved = {"t2v_gn_stats": 1, "t2v_gn_apply": 2, "t2v_gn_bwd_stats": 2, "t2v_gn_bwd_apply": 3 + (1 if addend else 0)}[name] * E * 2
                # algorithmic minimum: forward reads x once and writes y once (statistics from the same read);
                # backward reads x and dy once and writes dx (+ reads the residual gradient it absorbs)
                alg = {"t2v_gn_stats": 0, "t2v_gn_apply": 2, "t2v_gn_bwd_stats": 0, "t2v_gn_bwd_apply": 3 + (1 if addend else 0)}[name] * E * 2
            norms.append((kind, alg, moved, s, e, inner))
            if name == "t2v_layernorm_fwd" and tflag["temporal"] > 0:
                tunit["ln"].append((s, e, inner))
            return r
        if name == "t2v_lora_wgrad_batch":      # a[0] = array of descriptors, a[1] = their number: one launch for all of them
            fl = by = 0.0
            for i in range(a[1]):
                d = a[0][i]
                taps = d.geom.KH * d.geom.KW if d.conv else 1
                fl += 2.0 * d.rp * (d.N + taps * d.C) * d.rows
                by += (d.N + d.C) * d.rows * 2.0
            wgrad.append((fl, by, s, e, inner))
            wgrad_layers[0] += a[1]
            return r
        d = a[0]._obj
        if name == "t2v_lora_wgrad":      # each activation operand read once; fp32 factor gradients accumulated
            wgrad_layers[0] += 1
            taps = d.geom.KH * d.geom.KW if d.conv else 1
            wgrad.append((2.0 * d.rp * (d.N + taps * d.C) * d.rows, (d.N + d.C) * d.rows * 2.0, s, e, inner))
            return r
        bh = float(d.nbatch) * d.heads * 64
        fwd = name == "t2v_attn_fwd"
        flops = 4.0 * bh * d.Sq * d.Sk * (1.0 if fwd else 2.5)           # QK^T + PV; backward: 5 products
        nbytes = bh * (2 * d.Sq + 2 * d.Sk) * 2.0 * (1.0 if fwd else 2.0)   # q,k,v,o (+ do,dq,dk,dv)
        kind = "text_cross" if d.Sk == 77 else ("temporal" if d.Sq == d.Sk and d.Sq <= 64 else "spatial")
        attn.append((kind, flops, nbytes, s, e, inner))
        if fwd and tflag["attn"] > 0:
            tunit["core"].append((s, e, inner))
        return r

    _in_forward = [True]
    F.launch_gemm = timed
    F.launch_gemm_pair = timed_pair
    nv.call = timed_call
    _leaves.TransformerTemporalModel.forward = tt_forward
    _leaves.Attention.forward = at_forward
    side_was = F._side["enabled"]
    F._side["enabled"] = False        # one stream for the instrumented pass: an event pair on the side stream would also count
    try:                              # the time its kernel waits for the main stream's kernels to leave the CUs
        trainer.opt.zero_grad()
        trainer._fwd_bwd(batch)
        torch.cuda.synchronize()
    finally:
        F._side["enabled"] = side_was
        F.launch_gemm = orig
        F.launch_gemm_pair = orig_pair
        nv.call = orig_call
        _leaves.TransformerTemporalModel.forward = _tt_fwd
        _leaves.Attention.forward = _at_fwd
    out = {}
    def kernel_ms(r):
        return dur(r[1], r[2], r[5])

    for name, sel in (("nn", True), ("kmajor", False)):
        rs = [r for r in records if r[3] == sel]
        out[name] = dict(launches=len(rs), flops=sum(r[0] for r in rs), ms=sum(kernel_ms(r) for r in rs),
                         ms_event_pairs=sum(r[1].elapsed_time(r[2]) for r in rs),
                         kernel_timestamped=sum(1 for r in rs if r[5] is not None), abytes=sum(r[4] for r in rs))

    def both_roofs(flops, nbytes, ms, launches):
        t = ms * 1e-3
        return {"launches": launches, "ms_per_step": round(ms, 3), "TFLOP/s": round(flops / t / 1e12, 1),
                "frac_mfma_peak": round(flops / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "algorithmic_GB_per_step": round(nbytes / 1e9, 3),
                "GB/s": round(nbytes / t / 1e9, 1), "frac_hbm_peak": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)}

    ns = {}
    if conv3d:
        ns["conv3d_3x1x1"] = both_roofs(sum(c[0] for c in conv3d), sum(c[1] for c in conv3d),
                                        sum(dur(c[2], c[3], c[4]) for c in conv3d), len(conv3d))
        # the weight-streaming end of the same family (the only launches where "Conv3d against the HBM roof" is
        # coherent — M <= 512 rows, 3 C^2 weights dominate the bytes)
        small = [c for c in conv3d if c[5] <= 512]
        if small:
            ns["conv3d_3x1x1_small_M"] = both_roofs(sum(c[0] for c in small), sum(c[1] for c in small),
                                                    sum(dur(c[2], c[3], c[4]) for c in small), len(small))
            ns["conv3d_3x1x1_small_M"]["rows"] = sorted({int(c[5]) for c in small})
    if wgrad:
        ns["lora_factor_gradients"] = both_roofs(sum(c[0] for c in wgrad), sum(c[1] for c in wgrad),
                                                 sum(dur(c[2], c[3], c[4]) for c in wgrad), len(wgrad))
        ns["lora_factor_gradients"]["layers"] = wgrad_layers[0]
    for kind in ("temporal", "spatial", "text_cross"):
        rs = [r for r in attn if r[0] == kind]
        if rs:
            ns[f"{kind}_attention_core"] = both_roofs(sum(r[1] for r in rs), sum(r[2] for r in rs),
                                                      sum(dur(r[3], r[4], r[5]) for r in rs), len(rs))
    for kind in ("gn_fwd", "gn_bwd", "gn_finish", "ln_fwd", "ln_bwd"):
        rs = [r for r in norms if r[0] == kind]
        if rs:
            ms = sum(dur(r[3], r[4], r[5]) for r in rs)
            alg, moved = sum(r[1] for r in rs), sum(r[2] for r in rs)
            if kind == "gn_finish":
                ns["groupnorm_finish(statistics from GEMM-epilogue column sums, fwd+bwd)"] = {"launches": len(rs), "ms_per_step": round(ms, 3)}
                continue
            ns[{"gn_fwd": "groupnorm_fwd(stats+apply)", "gn_bwd": "groupnorm_bwd(stats+apply)", "ln_fwd": "layernorm_fwd",
                "ln_bwd": "layernorm_bwd"}[kind]] = {
                "launches": len(rs), "ms_per_step": round(ms, 3), "algorithmic_GB_per_step": round(alg / 1e9, 3),
                "GB/s": round(alg / ms / 1e6, 1), "frac_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                "moved_GB_per_step": round(moved / 1e9, 3), "moved_GB/s": round(moved / ms / 1e6, 1)}
    if tunit["calls"]:
        # two of the three LayerNorms of a temporal BasicTransformerBlock precede an attention (norm1, norm2)
        t_ms = (sum(dur(*r) for r in tunit["gemm"]) + sum(dur(*r) for r in tunit["core"]) + sum(dur(*r) for r in tunit["ln"]) * 2.0 / 3.0)
        ns["temporal_block_fused_unit"] = {
            "what": "SURVEY 8(d) fused unit (8 C^2 T + 4 T F C) / sum t over the forward launches of every temporal attention "
                    "(LayerNorm, QKV projection, FxF attention core, output projection + residual); the unit is not ONE kernel here",
            "units": tunit["calls"], "launches": len(tunit["gemm"]) + len(tunit["core"]) + int(len(tunit["ln"]) * 2 / 3),
            "ms_per_step": round(t_ms, 3), "ms_projections": round(sum(dur(*r) for r in tunit["gemm"]), 3),
            "ms_attention_core": round(sum(dur(*r) for r in tunit["core"]), 3),
            "ms_layernorm": round(sum(dur(*r) for r in tunit["ln"]) * 2.0 / 3.0, 3), "GFLOP_per_step": round(tunit["flops"] / 1e9, 1),
            "TFLOP/s": round(tunit["flops"] / t_ms / 1e9, 1),
            "frac_mfma_peak": round(tunit["flops"] / t_ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)}
    out["north_star"] = ns
    if os.environ.get("T2V_BENCH_SHAPE_TABLE"):      # per-problem-signature GEMM time of one step (diagnostic)
        agg = {}
        for key, fl, s, e, inner in shapes:
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1; a[1] += dur(s, e, inner); a[2] += fl
        tot = sum(a[1] for a in agg.values())
        with open(os.environ["T2V_BENCH_SHAPE_TABLE"], "w") as f:
            f.write("# per-signature GEMM time of one step (kernel timestamps where the hook applies): (M, N, rank cols, K, batch, "
                    "window, kernel, residual)\n")
            for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{str(key):72s} launches {n:4d}  ms {ms:8.3f} ({100 * ms / tot:4.1f} %)  us/launch {ms / n * 1e3:8.1f}  "
                        f"GFLOP {fl / n / 1e9:8.2f}  TF/s {fl / ms / 1e9:7.1f}\n")
    if kev is not None:
        kev.close()
    return out


def temporal_fused_forward(unet, frames, H, W, dev):
    """SURVEY 8(d)'s fused temporal unit as ONE kernel (csrc/temporal_fused.hip), measured where the product runs it: a no-grad,
    CFG-doubled UNet forward in eval mode = one UNet call of a sampling step (train.py:908-958, inference.py:153-267) on the
    benchmark's clip shape, LoRA wrappers folded (W + s up down).  Every `t2v_temporal_fused_fwd` launch is timed by its own
    begin / end timestamps (hipExtLaunchKernelGGL through the library's measurement hook); flops = 8 C^2 T + 4 T F C per unit.
    Beside it: the wall time of that forward with the one-launch unit and with the separate launches (T2V_TEMPORAL_FUSED=0 path)."""
    import t2v_amd.functional as F
    nv = F.nv
    lib = nv.lib()
    kev = _KernelEvents()
    recs = []
    orig_call = nv.call

    def timed_call(name, *a):
        if name != "t2v_temporal_fused_fwd":
            return orig_call(name, *a)
        d = a[0]._obj
        inner = kev.pair()
        if inner is not None:
            lib.t2v_launch_timing_events(inner[0], inner[1])
        r = orig_call(name, *a)
        if inner is not None and lib.t2v_launch_timing_consumed():
            recs.append((int(d.C), int(d.B) * int(d.F) * int(d.HW), int(d.F), inner))
        return r

    was_training = unet.training
    unet.eval()
    g = torch.Generator(device="cpu").manual_seed(99)
    lat = torch.randn(2, 4, frames, H // 8, W // 8, generator=g).to(dev)
    ts = torch.tensor([500, 500], device=dev)
    ehs = torch.randn(2, 77, 1024, generator=g).to(dev)

    def fwd():
        with torch.no_grad():
            return unet(lat, ts, encoder_hidden_states=ehs).sample

    def wall(n=5):
        fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fwd()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    try:
        out_f = fwd()
        ms_fused = wall()
        nv.call = timed_call
        fwd()
        torch.cuda.synchronize()
        nv.call = orig_call
        was = F._temporal_fused
        F._temporal_fused = False
        try:
            out_s = fwd()
            ms_sep = wall()
        finally:
            F._temporal_fused = was
        rel = float((out_f.float() - out_s.float()).norm() / out_s.float().norm())
        # the same call as the sampler issues it: captured once, replayed per timestep (pipelines.TextToVideoSampler)
        ms_graph = None
        try:
            fwd()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fwd()
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            ms_graph = (time.perf_counter() - t0) / 5 * 1e3
            del gr
        except Exception as e:   # noqa: BLE001
            print(f"[bench] capturing the sampling forward failed: {type(e).__name__}: {e}", file=sys.stderr)
    finally:
        nv.call = orig_call
        unet.train(was_training)
    if not recs:
        kev.close()
        return None
    by = {}
    for Cc, T, Fr, inner in recs:
        a = by.setdefault(Cc, [0, 0.0, 0.0, T])
        a[0] += 1
        a[1] += kev.ms(inner) or 0.0
        a[2] += 8.0 * Cc * Cc * T + 4.0 * T * Fr * Cc
    kev.close()
    fl, ms = sum(a[2] for a in by.values()), sum(a[1] for a in by.values())
    return {"what": "SURVEY 8(d) fused temporal unit LN -> QKV -> FxF softmax -> PV -> out-proj -> +residual as ONE kernel "
                    "(temporal_fused_fwd_kernel<C,NBO>), forward-only: the launches of one no-grad CFG-doubled UNet forward in eval mode "
                    "(a sampling step's UNet call) on this config's clip; kernel begin/end timestamps",
            "units_one_launch": len(recs), "units_total": 34, "ms": round(ms, 3), "GFLOP": round(fl / 1e9, 1),
            "TFLOP/s": round(fl / ms / 1e9, 1), "frac_mfma_peak": round(fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4),
            "per_width": {str(Cc): {"units": a[0], "rows": a[3], "us_per_unit": round(a[1] / a[0] * 1e3, 1),
                                    "TFLOP/s": round(a[2] / a[1] / 1e9, 1), "frac_mfma_peak": round(a[2] / a[1] / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)}
                          for Cc, a in sorted(by.items())},
            "policy": "widths <= 512 take the one-launch kernel (functional.temporal_fused_ok); the C = 1280 levels have 4 - 16 row tiles "
                      "per launch and stay on separate launches, C = 640 loses to them (registers)",
            "sampling_unet_forward_ms": {"graph_replay_one_launch_units": None if ms_graph is None else round(ms_graph, 2),
                                         "eager_one_launch_units": round(ms_fused, 2), "eager_separate_launches": round(ms_sep, 2),
                                         "note": "batch 2 (CFG pair), host-timed over 5 calls; the eager forms are bound by the host's launch path"},
            "one_launch_vs_separate_relerr": round(rel, 6)}


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel family in the step AS IT RUNS (every tile the shipped table selects),
    from the committed whole-step counter passes (scripts/pmc_step.sh -> profiles/r06_pmc_step.json, else an earlier round.s: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE, separate passes; PMC collection is slow and never part of the timed bench)."""
    for name in ("r06_pmc_step.json", "r05_pmc_step.json", "r04_pmc_step.json"):      # the newest committed counter passes
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                j = json.load(f)
            fam = next(v for k, v in j["families"].items() if k.startswith("gemm_kernel_dma"))
            return {"kernel_family": "gemm_kernel_dma<...> + gemm_w8_kernel<...> + gemm_skinny_kernel<MB>: all forward / backward-data launches of one C2 step",
                    "hbm_bytes_per_launch": fam["hbm_bytes_per_launch"], "hbm_GB_per_step": fam["hbm_GB_per_step"],
                    "launches_per_step": fam["launches_per_step"], "fetch_correction": j["fetch_correction"],
                    "source": "profiles/" + name, "note": j.get("note")}
        except Exception:   # noqa: BLE001
            continue
    return None


def rocprof_family_time(algorithmic_flops):
    """The same family's kernel time in the steady-state GRAPH REPLAY of this bench, from the committed rocprofv3 kernel trace
    (scripts/profile_bench.sh -> profiles/r06_bench_c2_kernel_stats.txt; pure kernel durations, no per-launch dispatch gap).
    The live event pairs of the eager instrumented pass above also contain each launch's dispatch latency (~5 us x 955), which a
    graph replay overlaps with the previous kernel; both numbers are reported."""
    try:
        ms = 0.0
        n = 0
        src = next(f for f in ("r06_bench_c2_kernel_stats.txt", "r05_bench_c2_kernel_stats.txt", "r04_bench_c2_kernel_stats.txt")
                   if os.path.exists(os.path.join(ROOT, "profiles", f)))      # the newest committed trace
        with open(os.path.join(ROOT, "profiles", src)) as f:
            for line in f:
                if ("gemm_kernel_dma" in line or "gemm_w8_kernel" in line or "gemm_skinny_kernel" in line) and line.lstrip().startswith("_Z"):
                    parts = line.split()
                    n += int(float(parts[-4]))
                    ms += float(parts[-3])
        if ms <= 0:
            return None
        tf = algorithmic_flops / (ms * 1e-3) / 1e12
        return {"kernel_ms_per_step": round(ms, 2), "launches_per_step": n, "TFLOP/s": round(tf, 1),
                "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "source": f"profiles/{src} (config c2)"}
    except Exception:   # noqa: BLE001
        return None


def cpu_baseline(steps, device, c2_once=True):
    """The CPU oracle (restatement of the reference path; the reference itself needs diffusers, absent offline) timed on
    the host cores: full train steps of config C1 (8 frames @128x128, LoRA r=4, batch 1), one untimed warm-up, then the
    MEDIAN of `steps` timed steps.  The first (warm-up) step doubles as the in-run parity check: the native trainer
    evaluates the eps-MSE of the same weights and inputs on the GPU -> `eps_mse_rel_err` (bar: 1e-3)."""
    import statistics
    from oracle.fastconv import fast_temporal_conv3d
    from oracle.lora import inject_trainable_lora_extended
    from oracle.train_step import finetune_unet_loss, train_step
    from oracle.unet3d import UNet3DConditionModel
    from oracle.vae import AutoencoderKLEncoder
    from oracle.weights import randomize_temporal_conv4, synthetic_batch as obatch
    frames, H, W, r = CONFIGS["c1"]
    torch.manual_seed(0)
    with torch.device(device):                                 # random init on the GPU only to make it fast
        unet = UNet3DConditionModel()
        vae = AutoencoderKLEncoder()
    unet, vae = unet.cpu(), vae.cpu().eval()
    randomize_temporal_conv4(unet)
    unet.requires_grad_(False)
    vae.requires_grad_(False)
    inject_trainable_lora_extended(unet, {"UNet3DConditionModel"}, r=r)
    from oracle.weights import randomize_lora_up
    randomize_lora_up(unet, scale=0.02)          # lora_up ~ N(0, 0.02^2): at the reference's init (up = 0) every LoRA branch is
    for m in unet.modules():                     # arithmetically inert and the in-run check would only see the frozen path
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    unet.train()
    cores = max(1, min(32, os.cpu_count() or 1))              # oneDNN stops scaling (and oversubscribes) beyond ~32 threads here
    torch.set_num_threads(cores)
    # ---- in-run parity: same weights, same host-drawn batch, native path on the GPU vs oracle on the CPU
    batch = obatch(frames, H, W, seed=1234)
    rel, l_cpu, l_gpu = None, None, None
    try:
        import t2v_amd  # noqa: F401
        from t2v_amd.models.unet_3d_condition import UNet3DConditionModel as DUNet
        from t2v_amd.models.vae import AutoencoderKL
        from t2v_amd.training import DenoiseTrainer
        from t2v_amd.utils.lora import inject_trainable_lora_extended as dinject
        with torch.device("meta"):
            dunet, dvae = DUNet(), AutoencoderKL()
        dunet, dvae = dunet.to_empty(device=device), dvae.to_empty(device=device)
        dunet.requires_grad_(False); dvae.requires_grad_(False)
        dinject(dunet, {"UNet3DConditionModel"}, r=r)
        dunet.load_state_dict(unet.state_dict(), strict=True); dvae.load_state_dict(vae.state_dict(), strict=True)
        for m in dunet.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        dunet.train()
        tr = DenoiseTrainer(dunet, dvae.eval(), [p for p in dunet.parameters() if p.requires_grad], lr=5e-6)
        with torch.no_grad():
            l_gpu = float(tr.loss_fn({k: v.to(device) for k, v in batch.items()}))
        del tr, dunet, dvae
        torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001
        print(f"[bench] in-run parity check failed: {type(e).__name__}: {e}", file=sys.stderr)
    params = [p for p in unet.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=5e-6, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    times = []
    with fast_temporal_conv3d():      # (3,1,1) Conv3d evaluated as a (3,1) conv2d: same arithmetic, oneDNN fast path
        # warm-up step (untimed) = the parity reference: loss BEFORE any update
        lw, _ = finetune_unet_loss(unet, vae, batch)
        l_cpu = float(lw.detach())
        lw.backward()
        opt.step(); opt.zero_grad(set_to_none=True)
        for i in range(steps):
            b = obatch(frames, H, W, seed=2000 + i)
            t0 = time.time()
            train_step(unet, vae, b, opt)
            times.append(time.time() - t0)
        c2_s = None
        if c2_once:             # the C2 clip once (same oracle objects: LoRA r=4 instead of 16 changes < 0.1 % of the work)
            f2, H2, W2, _ = CONFIGS["c2"]
            b2 = obatch(f2, H2, W2, seed=3000)
            t0 = time.time()
            train_step(unet, vae, b2, opt)
            c2_s = time.time() - t0
    if l_gpu is not None:
        rel = abs(l_gpu - l_cpu) / abs(l_cpu)
    med = statistics.median(times)
    return dict(value=1.0 / med, unit="videos/s", cores=cores, kind="port",
                sample=f"median of {steps} full train steps of config C1 (8 frames @128x128, LoRA r=4, fp32, PyTorch CPU oracle on "
                       f"{cores} threads, temporal Conv3d run as conv2d) after one warm-up step: {med:.2f} s/step "
                       f"(all: {', '.join(f'{t:.1f}' for t in times)})"
                       + (f"; ONE full train step at the C2 clip (16 frames @256x256) on the same threads: {c2_s:.1f} s = "
                          f"{1.0 / c2_s:.4f} videos/s" if c2_s else "; a C1 clip is ~1/8 of the C2 clip's work"),
                c2_step_s=(round(c2_s, 2) if c2_s else None), c2_videos_per_s=(round(1.0 / c2_s, 5) if c2_s else None),
                eps_mse_rel_err=rel, eps_mse_cpu=l_cpu, eps_mse_gpu=l_gpu,
                eps_mse_note="same seeded ModelScope-1.7B weights (lora_up ~ N(0, 0.02^2): live LoRA branches) and C1 batch: native "
                             "trainer loss on the GPU vs the CPU fp32 oracle, this run; full-size C1/C2 loss + every factor gradient "
                             "against committed oracle fixtures: tests/test_zz_fullsize_gpu.py")


def _dist_info(world):
    """Backend, collective-library version and the world size `torch.distributed` itself reports (N > 1 lines)."""
    info = {"world_size_env": world, "initialized": False}
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            info.update(initialized=True, backend=dist.get_backend(), world_size=dist.get_world_size(), rank=dist.get_rank())
            if dist.get_backend() == "nccl":
                v = torch.cuda.nccl.version()
                info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        info["exchange"] = ("one all_reduce(SUM) of the flat fp32 gradient buffer per optimiser step (tail slot = the loss), "
                            "stream-ordered behind the step graph" if world > 1 else "none (one GPU)")
    except Exception as e:   # noqa: BLE001
        info["error"] = f"{type(e).__name__}: {e}"
    return info


def _self_spawn(args):
    """`python bench.py --gpus N` without a torchrun environment: re-launch under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_spawn(args)
    import t2v_amd  # noqa: F401
    from t2v_amd.parallel import init_from_env
    from t2v_amd.training import DenoiseTrainer
    # (T2V_BENCH_BACKEND=gloo + T2V_BENCH_SHARE_GPU=1: two ranks on ONE device over gloo — the plumbing test of tests/test_dp_gpu.py;
    #  RCCL refuses two ranks on one GPU)
    rank, world, local = init_from_env(os.environ.get("T2V_BENCH_BACKEND", "nccl") if args.gpus > 1 else None)
    if os.environ.get("T2V_BENCH_SHARE_GPU") == "1":
        local = local % max(1, torch.cuda.device_count())
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU path in the product)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    frames, H, W, r = CONFIGS[args.config]
    args.dropout = not args.eval_train          # headline = the mode train.py drives by default (train.py:779: eval_train=False)

    unet, vae, trainable = build_models(frames, r, dev, seed=0, dropout=args.dropout,
                                        grad_ckpt=args.grad_checkpointing)   # same frozen weights on every rank
    text_encoder = None if args.no_text_encoder else build_text_encoder(dev)
    trainer = DenoiseTrainer(unet, vae, trainable, lr=5e-6, world_size=world, text_encoder=text_encoder)
    if world > 1:
        from t2v_amd.parallel import broadcast_params
        broadcast_params(trainer.opt.flat_p)
    batch = synthetic_batch(frames, H, W, dev, seed=1234 + rank, with_ids=text_encoder is not None)   # one clip per GPU
    if args.config in CACHED_LATENTS:        # train.py:741-746: the dataset hands over latents, no VAE encode in the step
        g = torch.Generator(device="cpu").manual_seed(4321 + rank)
        batch["latents"] = (torch.randn(1, 4, frames, H // 8, W // 8, generator=g) * 0.18215 * 4.0).to(dev)
        del batch["pixel_values"]

    use_graph = not args.no_graph             # active dropout is captured too: the device-side dropout epoch moves the masks
    text_mode = "clip-in-step" if text_encoder is not None else "synthetic"
    if use_graph:
        try:
            trainer.capture(batch, warmup=1)
        except Exception as e:   # noqa: BLE001  (a transformers op that cannot be captured: keep CLIP eager, outside the graph)
            if text_encoder is None:
                raise
            print(f"[bench] capturing CLIP failed ({type(e).__name__}); running it eagerly before each replay", file=sys.stderr)
            torch.cuda.synchronize()
            text_mode = "clip-eager-before-replay"
            ids = batch.pop("prompt_ids")
            with torch.no_grad():
                batch["encoder_hidden_states"] = text_encoder(ids[0])[0]
            trainer.capture(batch, warmup=1, pipelined=False)
            _replay = trainer.replay_step

            def step():
                with torch.no_grad():
                    trainer._static["encoder_hidden_states"].copy_(text_encoder(ids[0])[0])
                return _replay()
        else:
            step = lambda: trainer.replay_step()
    else:
        step = lambda: trainer.train_step(batch)
    for _ in range(args.warmup):
        loss = step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    # host time to ISSUE one step (zero_grad, graph launch, exchange, clip + AdamW launches) with the device idle: what bounds
    # the step rate at N > 1 once the device work shrinks (a replay of a still-running graph blocks the host, so the loop above
    # cannot show it); median of 5
    hs = []
    for _ in range(0 if args.no_host_timing else 5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = step()
        hs.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_ms = round(sorted(hs)[2] * 1e3, 3) if hs else None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss.item())
    trainer.check_device_flags()        # a split-K hand-off that gave up inside the timed region voids the line: fail loudly

    default_mode_ms = None
    if (rank == 0 and world == 1 and args.config == "c2" and not args.no_default_mode and use_graph):
        # the OTHER dropout mode on the same clip, a short graph-replayed run beside the headline: the reference's opt-in
        # eval_train mode (train.py:779-781, every Dropout off -> LoRA branches merged into the weights) beside the default-mode
        # headline, or the default mode beside an --eval-train line
        try:
            d_unet, d_vae, d_train = build_models(frames, r, dev, seed=0, dropout=not args.dropout, grad_ckpt=args.grad_checkpointing)
            d_tr = DenoiseTrainer(d_unet, d_vae, d_train, lr=5e-6, world_size=1, text_encoder=text_encoder)
            d_batch = synthetic_batch(frames, H, W, dev, seed=1234, with_ids=text_encoder is not None)
            d_tr.capture(d_batch, warmup=1)
            for _ in range(2):
                d_tr.replay_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                d_tr.replay_step()
            torch.cuda.synchronize()
            default_mode_ms = round((time.perf_counter() - t1) / 10 * 1e3, 2)
            del d_tr, d_unet, d_vae, d_train
            torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            print(f"[bench] default-mode (dropout) run failed: {type(e).__name__}: {e}", file=sys.stderr)

    roof = None
    if rank == 0 and world == 1 and not args.no_roofline:      # per-kernel instrumentation belongs to the 1-GPU line
        both = gemm_roofline(trainer, batch)
        rr, km = both["nn"], both["kmajor"]
        ach = rr["flops"] / (rr["ms"] * 1e-3) / 1e12
        roof = dict(bound="mfma",
                    kernel="gemm_w8_kernel<BM,BN,WM,WN,KG,NSTAGE,SCHED,BK,CS> + gemm_kernel_dma<BM,BN,WM,WN,NSTAGE> + gemm_skinny_kernel<MB> (M <= 96: the CLIP tower) - every Linear/Conv2d/"
                           "Conv3d forward and backward-data launch of one step (implicit GEMM, LDS-DMA ring; 8-wave and 4-wave "
                           "families, tile per signature from the shipped table); eager instrumented pass, HIP events on the launch stream: kernel begin/end "
                           "timestamps (hipExtLaunchKernelGGL) for single-kernel launches, an event pair around the call for split-K pairs",
                    achieved=round(ach, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    launches=rr["launches"], algorithmic_gflop_per_step=round(rr["flops"] / 1e9, 1),
                    kernel_ms_per_step=round(rr["ms"], 2), event_pair_ms_per_step=round(rr["ms_event_pairs"], 2),
                    launches_kernel_timestamped=rr["kernel_timestamped"], traffic=(pmc_traffic() or {}).get("hbm_bytes_per_launch"),
                    traffic_source=((pmc_traffic() or {}).get("source", "") + " (separate rocprofv3 --pmc passes over this bench, not this run)") or None,
                    algorithmic_bytes_per_launch=int(rr["abytes"] / max(1, rr["launches"])),
                    algorithmic_GB_per_step=round(rr["abytes"] / 1e9, 2), traffic_detail=pmc_traffic(),
                    secondary={"kernel": "gemm_kernel<..,AT|BT> - K-major / transposed-operand launches (factor gradients of strided convs, "
                                         "VAE attention P.V, and at c3 the full weight gradients); the LoRA factor gradients proper are "
                                         "north_star_kernels.lora_factor_gradients",
                               "launches": km["launches"], "algorithmic_gflop_per_step": round(km["flops"] / 1e9, 1),
                               "kernel_ms_per_step": round(km["ms"], 2), "bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "achieved": round(km["flops"] / max(km["ms"], 1e-9) / 1e9, 1),
                               "frac": round(km["flops"] / max(km["ms"], 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4),
                               "note": "at config c3 (full finetune) this family carries every full weight gradient dW = x^T dy"},
                    graph_replay_rocprof=rocprof_family_time(rr["flops"]) if args.config == "c2" else None,
                    north_star_kernels=both["north_star"])
        try:
            tfu = temporal_fused_forward(unet, frames, H, W, dev)
            if tfu is not None:
                roof["north_star_kernels"]["temporal_fused_forward_unit"] = tfu
        except Exception as e:   # noqa: BLE001
            print(f"[bench] fused temporal unit measurement failed: {type(e).__name__}: {e}", file=sys.stderr)
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_steps, dev, c2_once=not args.no_cpu_c2)

    if rank == 0:
        out = {
            "metric": METRICS[args.config],
            "value": round(args.steps * world / dt, 4), "unit": "videos/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: ModelScope-1.7B UNet3D + "
                                   + ("cached latents (no VAE in the step)" if args.config in CACHED_LATENTS else "SD-VAE encode")
                                   + f", {frames} frames @{W}x{H}, "
                                   + (f"LoRA r={r} on all 574 Linear/Conv layers" if r > 0 else "full UNet finetune (no LoRA)")
                                   + f", batch 1 clip/GPU, 2 UNet passes/step, "
                                   + ("LoRA dropout 0.1 + TemporalConvLayer dropout 0.1 (reference default train mode)" if args.dropout
                                      else "dropout off (reference eval_train mode)")
                                   + (", gradient checkpointing on" if args.grad_checkpointing else "")
                                   + f", text encoder: {text_mode}",
                       "global_batch": world, "parallelism": f"dp{world}", "graph_replay": use_graph,
                       "graph_pipeline": bool(use_graph and getattr(trainer, "_pipe", None) is not None),
                       "trainable_params": sum(p.numel() for p in trainer.opt.params),
                       "flat_gradient_elems": trainer.opt.numel, "final_loss": final_loss,
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                       "eps_mse_rel_err": (cpu or {}).get("eps_mse_rel_err"),
                       "default_mode_ms_per_step": (round(dt / args.steps * 1e3, 2) if args.dropout else default_mode_ms),
                       "eval_train_ms_per_step": (default_mode_ms if args.dropout else round(dt / args.steps * 1e3, 2)),
                       "mode_note": "headline = the reference's DEFAULT train mode (LoRA dropout 0.1 + TemporalConvLayer dropout 0.1, masks "
                                    "restated in oracle/dropout.py); eval_train_ms_per_step = the same clip with every Dropout off "
                                    "(train.py:779-781 opt-in), 10 graph replays" if args.dropout else
                                    "--eval-train line; default_mode_ms_per_step = the same clip in the reference's default mode, 10 graph replays",
                       "host_ms_per_step": host_ms,
                       # what the exchange ran on: lets an N > 1 line be audited
                       "dist": _dist_info(world)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if args.export_tune_table and rank == 0:
        import t2v_amd.native as nv
        print(f"[bench] exported {nv.export_tune_table(args.export_tune_table)} tile-table entries to {args.export_tune_table}",
              file=sys.stderr)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
