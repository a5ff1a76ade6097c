"""One UNet call of a sampling step (CFG pair, no grad, eval mode, LoRA wrappers in place) at a config's clip shape: captured once,
replayed `reps` times — what pipelines.TextToVideoSampler does per timestep.  Prints ms per replay; run it under
`scripts/profile_cmd.sh <tag> scripts/sampling_probe.py [c2|c4]` for the per-kernel statistics of the replays.
    python scripts/sampling_probe.py [c2|c4] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frames, H, W, r = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
unet, vae, _ = bench.build_models(frames, r, dev, seed=0, dropout=True)
for n_, p_ in unet.named_parameters():          # live branches (the reference's init leaves lora_up at zero)
    if "lora_up" in n_:
        torch.nn.init.normal_(p_, std=0.02)
unet.eval()
g = torch.Generator(device="cpu").manual_seed(99)
lat = torch.randn(2, 4, frames, H // 8, W // 8, generator=g).to(dev)
ts = torch.tensor([500, 500], device=dev)
ehs = torch.randn(2, 77, 1024, generator=g).to(dev)


def fwd():
    with torch.no_grad():
        return unet(lat, ts, encoder_hidden_states=ehs).sample


for _ in range(2):
    fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    fwd()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 3 * 1e3
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = fwd()
gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    gr.replay()
torch.cuda.synchronize()
print(f"{cfg}: CFG-pair UNet call ({frames} frames @{W}x{H}, LoRA r={r} folded): eager {eager:.2f} ms, graph replay {(time.perf_counter() - t0) / reps * 1e3:.2f} ms "
      f"(finite: {bool(torch.isfinite(out).all())})", flush=True)
