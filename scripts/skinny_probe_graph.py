"""Kernel time of the skinny dense kernels on the CLIP tower's four layer shapes, 100 launches replayed from a HIP graph (the
eager call rate is dispatch-bound at ~28 us).  T2V_GEMM_SKINNY=1|2|3 python scripts/skinny_probe_graph.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import t2v_amd  # noqa: F401
import t2v_amd.functional as F

mode = os.environ.get("T2V_GEMM_SKINNY", "1")
out = []
for M, N, K in ((77, 3072, 1024), (77, 1024, 1024), (77, 4096, 1024), (77, 1024, 4096)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").bfloat16()
    st = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(st):
        for _ in range(3):
            F.conv_linear(x, w, b, residual=r)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(100):
                y = F.conv_linear(x, w, b, residual=r)
        g.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
    out.append(f"N={N} K={K}: {s.elapsed_time(e) * 10:.1f} us")
print(f"SKINNY={mode} graph-replayed launch time: " + "; ".join(out))
