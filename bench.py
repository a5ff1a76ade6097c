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
    # name: (frames, height, width, lora_rank)
    "c2": (16, 256, 256, 16),     # BASELINE.json configs[1] — the configuration the metric is quoted on
    "c1": (8, 128, 128, 4),       # configs[0] — the reference's CPU-runnable case
}
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E peak, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-text-encoder", action="store_true", help="feed synthetic text states instead of running CLIP")
    ap.add_argument("--export-tune-table", default=None,
                    help="write the GEMM tile table after the run (use with T2V_GEMM_AUTOTUNE=live; scripts/tune_gemm_table.sh)")
    return ap.parse_args()


def build_models(frames, lora_rank, device, seed):
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
    unet.requires_grad_(False)
    vae.requires_grad_(False)
    handler = LoraHandler(use_unet_lora=True)
    params, _ = handler.add_lora_to_model(True, unet, ["UNet3DConditionModel"], 0.0, None, r=lora_rank)
    unet.train()
    for m in unet.modules():                            # the reference's `eval_train` mode (train.py:779-781): dropout off
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    vae.eval()
    trainable = [p for p in unet.parameters() if p.requires_grad]
    return unet, vae, trainable


def build_text_encoder(device):
    """Random-init CLIP text tower in the ModelScope/OpenCLIP ViT-H shape (23 layers, d=1024, 16 heads, MLP 4096, 77 tokens;
    SURVEY.md A.10), frozen, bf16.  It is adjacent to the hot path (§8f row 2, ~45 GFLOP of a 24 TFLOP step) and runs through
    stock PyTorch-ROCm ops; it is inside the timed step so that no part of the reference's step is skipped."""
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


def gemm_roofline(trainer, batch):
    """One instrumented eager step: HIP events around every GEMM-family launch on the launch stream."""
    import t2v_amd.functional as F
    records = []
    orig = F.launch_gemm

    def timed(**kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig(**kw)
        e.record()
        z = max(1, kw.get("batch", 1))
        flops = 2.0 * kw["M"] * kw["N"] * kw["K"] * z
        geom = kw.get("geom")
        if geom is not None and geom.tdiv == 2:
            flops /= 4.0                                 # 3/4 of the gathered taps are structural zeros
        nn_kernel = not kw.get("a_trans", 0) and not kw.get("b_trans", 0)
        records.append((flops, s, e, nn_kernel))
        shapes.append(((kw["M"], kw["N"], kw["K"], z, "conv" if geom is not None else "lin",
                        "nn" if nn_kernel else "tn"), flops, s, e))
        if nn_kernel and geom is not None and geom.KH == 3 and geom.KW == 1:
            # the (3,1,1) Conv3d launches (forward + backward-data): SURVEY 8(d) bytes = x once + y once + weights once
            taps = 3
            conv3d.append((flops, (kw["M"] * (kw["K"] // taps) + kw["M"] * kw["N"] + kw["N"] * kw["K"]) * 2.0, s, e))

    orig_pair = F.launch_gemm_pair

    def timed_pair(kw_a, kw_b):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        orig_pair(kw_a, kw_b)
        e.record()
        records.append((2.0 * (kw_a["M"] * kw_a["N"] * kw_a["K"] + kw_b["M"] * kw_b["N"] * kw_b["K"]), s, e, False))

    conv3d, attn, wgrad, shapes = [], [], [], []
    nv = F.nv
    orig_call = nv.call

    def timed_call(name, *a):
        if name not in ("t2v_attn_fwd", "t2v_attn_bwd", "t2v_lora_wgrad"):
            return orig_call(name, *a)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig_call(name, *a)
        e.record()
        d = a[0]._obj
        if name == "t2v_lora_wgrad":      # each activation operand read once; fp32 factor gradients accumulated
            taps = d.geom.KH * d.geom.KW if d.conv else 1
            wgrad.append((2.0 * d.rp * (d.N + taps * d.C) * d.rows, (d.N + d.C) * d.rows * 2.0, s, e))
            return r
        bh = float(d.nbatch) * d.heads * 64
        fwd = name == "t2v_attn_fwd"
        flops = 4.0 * bh * d.Sq * d.Sk * (1.0 if fwd else 2.5)           # QK^T + PV; backward: 5 products
        nbytes = bh * (2 * d.Sq + 2 * d.Sk) * 2.0 * (1.0 if fwd else 2.0)   # q,k,v,o (+ do,dq,dk,dv)
        kind = "text_cross" if d.Sk == 77 else ("temporal" if d.Sq == d.Sk and d.Sq <= 64 else "spatial")
        attn.append((kind, flops, nbytes, s, e))
        return r

    F.launch_gemm = timed
    F.launch_gemm_pair = timed_pair
    nv.call = timed_call
    try:
        trainer.opt.zero_grad()
        trainer._fwd_bwd(batch)
        torch.cuda.synchronize()
    finally:
        F.launch_gemm = orig
        F.launch_gemm_pair = orig_pair
        nv.call = orig_call
    out = {}
    for name, sel in (("nn", True), ("kmajor", False)):
        rs = [r for r in records if r[3] == sel]
        out[name] = dict(launches=len(rs), flops=sum(r[0] for r in rs), ms=sum(r[1].elapsed_time(r[2]) for r in rs))

    def both_roofs(flops, nbytes, ms, launches):
        t = ms * 1e-3
        return {"launches": launches, "ms_per_step": round(ms, 3), "TFLOP/s": round(flops / t / 1e12, 1),
                "frac_mfma_peak": round(flops / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "algorithmic_GB_per_step": round(nbytes / 1e9, 3),
                "GB/s": round(nbytes / t / 1e9, 1), "frac_hbm_peak": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4)}

    ns = {}
    if conv3d:
        ns["conv3d_3x1x1"] = both_roofs(sum(c[0] for c in conv3d), sum(c[1] for c in conv3d),
                                        sum(c[2].elapsed_time(c[3]) for c in conv3d), len(conv3d))
    if wgrad:
        ns["lora_factor_gradients"] = both_roofs(sum(c[0] for c in wgrad), sum(c[1] for c in wgrad),
                                                 sum(c[2].elapsed_time(c[3]) for c in wgrad), len(wgrad))
    for kind in ("temporal", "spatial", "text_cross"):
        rs = [r for r in attn if r[0] == kind]
        if rs:
            ns[f"{kind}_attention_core"] = both_roofs(sum(r[1] for r in rs), sum(r[2] for r in rs),
                                                      sum(r[3].elapsed_time(r[4]) for r in rs), len(rs))
    out["north_star"] = ns
    if os.environ.get("T2V_BENCH_SHAPE_TABLE"):      # per-problem-signature GEMM time of one step (diagnostic)
        agg = {}
        for key, fl, s, e in shapes:
            a = agg.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl
        with open(os.environ["T2V_BENCH_SHAPE_TABLE"], "w") as f:
            for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{str(key):60s} launches {n:4d}  ms {ms:8.3f}  us/launch {ms / n * 1e3:8.1f}  TF/s {fl / ms / 1e9:7.1f}\n")
    return out


def pmc_traffic():
    """HBM bytes per launch of the family's heaviest single shape, from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_gemm_conv_l0.json; PMC collection is a separate, slow run — never part of the timed bench)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_gemm_conv_l0.json")) as f:
            j = json.load(f)
        return {"shape": "conv3x3 M=32768 N=320 K=2880 (stacked passes, 32x32 level)", "hbm_bytes_per_launch": j["hbm_bytes_per_launch"],
                "algorithmic_bytes_per_launch": j["algorithmic_bytes_per_launch"], "source": "profiles/r01_pmc_gemm_conv_l0.json"}
    except Exception:   # noqa: BLE001
        return None


def cpu_baseline(steps):
    """The CPU oracle (restatement of the reference path; the reference itself needs diffusers, absent offline)
    timed on the host cores: full train step of config C1 (8 frames @128x128, LoRA r=4, batch 1)."""
    from oracle.lora import inject_trainable_lora_extended
    from oracle.train_step import train_step
    from oracle.unet3d import UNet3DConditionModel
    from oracle.vae import AutoencoderKLEncoder
    from oracle.weights import randomize_temporal_conv4, synthetic_batch as obatch
    frames, H, W, r = CONFIGS["c1"]
    torch.manual_seed(0)
    dev = "cuda" if torch.cuda.is_available() else "cpu"      # build on the GPU only to make random init fast
    with torch.device(dev):
        unet = UNet3DConditionModel()
        vae = AutoencoderKLEncoder()
    unet, vae = unet.cpu(), vae.cpu().eval()
    randomize_temporal_conv4(unet)
    unet.requires_grad_(False)
    vae.requires_grad_(False)
    inject_trainable_lora_extended(unet, {"UNet3DConditionModel"}, r=r)
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    unet.train()
    params = [p for p in unet.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=5e-6, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    from oracle.fastconv import fast_temporal_conv3d
    cores = torch.get_num_threads()
    times = []
    with fast_temporal_conv3d():      # (3,1,1) Conv3d evaluated as a (3,1) conv2d: same arithmetic, oneDNN fast path
        for i in range(steps):
            batch = obatch(frames, H, W, seed=1234 + i)
            t0 = time.time()
            train_step(unet, vae, batch, opt)
            times.append(time.time() - t0)
    best = min(times)
    return dict(value=1.0 / best, unit="videos/s", cores=cores, kind="port",
                sample=f"{steps} full train step(s) of config C1 (8 frames @128x128, LoRA r=4, fp32, PyTorch CPU oracle, "
                       f"temporal Conv3d run as conv2d), best step {best:.2f} s; a C1 clip is ~1/8 of the C2 clip's work")


def main():
    args = parse()
    import t2v_amd  # noqa: F401
    from t2v_amd.parallel import init_from_env
    from t2v_amd.training import DenoiseTrainer
    rank, world, local = init_from_env("nccl" if args.gpus > 1 else None)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU path in the product)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    frames, H, W, r = CONFIGS[args.config]

    unet, vae, trainable = build_models(frames, r, dev, seed=0)            # same frozen weights on every rank
    text_encoder = None if args.no_text_encoder else build_text_encoder(dev)
    trainer = DenoiseTrainer(unet, vae, trainable, lr=5e-6, world_size=world, text_encoder=text_encoder)
    if world > 1:
        from t2v_amd.parallel import broadcast_params
        broadcast_params(trainer.opt.flat_p)
    batch = synthetic_batch(frames, H, W, dev, seed=1234 + rank, with_ids=text_encoder is not None)   # one clip per GPU

    use_graph = not args.no_graph
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
            trainer.capture(batch, warmup=1)
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
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss.item())

    roof = None
    if rank == 0 and world == 1 and not args.no_roofline:      # per-kernel instrumentation belongs to the 1-GPU line
        both = gemm_roofline(trainer, batch)
        rr, km = both["nn"], both["kmajor"]
        ach = rr["flops"] / (rr["ms"] * 1e-3) / 1e12
        roof = dict(bound="mfma",
                    kernel="gemm_kernel_dma<BM,BN,WM,WN,NSTAGE> - every Linear/Conv2d/Conv3d forward and backward-data launch of one "
                           "step (implicit-GEMM, LDS-DMA ring); eager instrumented pass, HIP events on the launch stream",
                    achieved=round(ach, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    launches=rr["launches"], algorithmic_gflop_per_step=round(rr["flops"] / 1e9, 1),
                    kernel_ms_per_step=round(rr["ms"], 2), traffic=(pmc_traffic() or {}).get("hbm_bytes_per_launch"),
                    traffic_detail=pmc_traffic(),
                    secondary={"kernel": "gemm_kernel<..,AT|BT> - K-major / transposed-operand launches (factor gradients of strided convs, "
                                         "VAE attention P.V); the LoRA factor gradients proper are north_star_kernels.lora_factor_gradients",
                               "launches": km["launches"], "algorithmic_gflop_per_step": round(km["flops"] / 1e9, 1),
                               "kernel_ms_per_step": round(km["ms"], 2)},
                    north_star_kernels=both["north_star"])
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_steps)

    if rank == 0:
        out = {
            "metric": "train-step videos/sec (16-frame 256x256, ModelScope-1.7B LoRA)" if args.config == "c2"
                      else "train-step videos/sec (8-frame 128x128, ModelScope-1.7B LoRA r=4)",
            "value": round(args.steps * world / dt, 4), "unit": "videos/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: ModelScope-1.7B UNet3D + SD-VAE encode, {frames} frames @{H}x{W}, "
                                   f"LoRA r={r} on all 574 Linear/Conv layers, batch 1 clip/GPU, 2 UNet passes/step, "
                                   f"dropout off (reference eval_train mode), text encoder: {text_mode}",
                       "global_batch": world, "parallelism": f"dp{world}", "graph_replay": use_graph,
                       "trainable_params": sum(p.numel() for p in trainer.opt.params),
                       "flat_gradient_elems": trainer.opt.numel, "final_loss": final_loss},
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
