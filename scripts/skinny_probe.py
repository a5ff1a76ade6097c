"""Parity + timing of the skinny dense kernels on the CLIP tower's four layer shapes (T2V_GEMM_SKINNY=1: four waves, =2: eight
waves with two chunks in flight, =0: the tiled kernels).  python scripts/skinny_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import t2v_amd  # noqa: F401
import t2v_amd.functional as F

torch.manual_seed(0)
mode = os.environ.get("T2V_GEMM_SKINNY", "1")
for M, N, K in ((77, 3072, 1024), (77, 1024, 1024), (77, 4096, 1024), (77, 1024, 4096), (33, 1000, 128)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16()
    with torch.no_grad():
        y = F.conv_linear(x, w, b, residual=r)
        ref = x.float() @ w.bfloat16().float().t() + b + r.float()
        err = float((y.float() - ref).norm() / ref.norm())
        for _ in range(5):
            F.conv_linear(x, w, b, residual=r)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(200):
            F.conv_linear(x, w, b, residual=r)
        e.record()
        torch.cuda.synchronize()
    print(f"SKINNY={mode} M={M} N={N} K={K}: rel err {err:.2e}, {s.elapsed_time(e) / 200 * 1e3:.1f} us per call (eager, incl. dispatch gap)")
F.check_gemm_workspaces()
print(f"SKINNY={mode}: no split-K give-up flag")
