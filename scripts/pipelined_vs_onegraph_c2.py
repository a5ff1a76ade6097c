"""the pipelined two-graph capture is the default on the evidence of a toy-size test.  This is the same check at the
benchmark configuration: three trainers on identical ModelScope-1.7B weights (C2: 16 frames @256x256, LoRA r=16; every Dropout off and
noise / timesteps / VAE sample handed over with the batch, so that the three forms see the same random inputs whatever their
warm-up consumed), one captured as prepare / UNet graph pairs, one as ONE single-stream graph, one eager; `steps` steps each on the
same batch sequence; prints the loss trajectories and their relative differences.  Expectation: step 0 equal to the fp32 atomics of the factor gradients (the loss itself is bit-reproducible since round
6), later steps within the run-to-run noise of two EAGER runs (profiles/r06_determinism.txt; lr 1e-5 keeps AdamW's sign-like first
updates from amplifying 1-ulp gradient differences).   python scripts/pipelined_vs_onegraph_c2.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2v_amd.training import DenoiseTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
frames, H, W, r = bench.CONFIGS["c2"]
dev = torch.device("cuda", 0)
losses = {}
for form in ("pipelined", "one_graph", "eager"):
    unet, vae, trainable = bench.build_models(frames, r, dev, seed=0, dropout=False)
    tr = DenoiseTrainer(unet, vae, trainable, lr=1e-5)
    batches = []
    for i in range(steps):
        b = bench.synthetic_batch(frames, H, W, dev, seed=100 + i)
        g = torch.Generator(device="cpu").manual_seed(900 + i)
        b["vae_eps"] = torch.randn(frames, 4, H // 8, W // 8, generator=g).to(dev)
        b["noise"] = torch.randn(1, 4, frames, H // 8, W // 8, generator=g).to(dev)
        b["timesteps"] = torch.randint(0, 1000, (1,), generator=g).to(dev)
        batches.append(b)
    if form != "eager":
        tr.capture(batches[0], warmup=1, pipelined=(form == "pipelined"))
        ls = [float(tr.replay_step(b)) for b in batches]
    else:
        ls = [float(tr.train_step(b)) for b in batches]      # (capture()'s warm-up passes restore the parameters they touched)
    torch.cuda.synchronize()
    tr.check_device_flags()
    losses[form] = ls
    print(f"{form:10s}: " + " ".join(f"{l:.6f}" for l in ls), flush=True)
    del tr, unet, vae, trainable
    torch.cuda.empty_cache()
for a, b in (("pipelined", "one_graph"), ("pipelined", "eager"), ("one_graph", "eager")):
    rel = [abs(x - y) / abs(y) for x, y in zip(losses[a], losses[b])]
    print(f"{a} vs {b}: max relative loss difference over {steps} steps {max(rel):.2e}  (per step: " + " ".join(f"{v:.1e}" for v in rel) + ")")
