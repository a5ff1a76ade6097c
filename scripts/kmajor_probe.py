"""Probe of the K-major (weight-gradient) GEMM: dW[N_out, K_in] = dy^T x over M tokens, on the C3 signatures.
    python scripts/kmajor_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import t2v_amd  # noqa: E402,F401
import t2v_amd.functional as F  # noqa: E402
import t2v_amd.native as nv  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for tokens, cout, cin in ((32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (2048, 10240, 1280), (32768, 2560, 320)):
    dy = torch.randn(tokens, cout, device=dev).to(torch.bfloat16)
    x = torch.randn(tokens, cin, device=dev).to(torch.bfloat16)
    dw = torch.zeros(cout, cin, dtype=torch.float32, device=dev)
    ref = dy.float().t() @ x.float()
    fl = 2.0 * tokens * cout * cin
    tiles64 = ((cout + 63) // 64) * ((cin + 63) // 64)
    for split in sorted({1, F._split_k(tiles64, tokens), 4, 16, 64}):
        if tokens // split < 128:
            continue
        for force in (os.environ.get("T2V_GEMM_FORCE_TILE", "heuristic"),):     # (read once per process: one run per tile)

            def run():
                F.launch_gemm(M=cout, N=cin, K=tokens, A=dy.data_ptr(), lda=cout, a_trans=1, B=x.data_ptr(), ldb=cin, b_trans=1,
                              D=dw.data_ptr(), ldd=cin, out_mode=nv.OUT_F32_ATOMIC, alpha=1.0, split_k=split)
            dw.zero_()
            try:
                run()
            except Exception as ex:   # noqa: BLE001
                print(f"   tokens {tokens} {cout}x{cin} split {split} tile {force}: {ex}")
                continue
            torch.cuda.synchronize()
            err = float((dw - ref).abs().max() / ref.abs().max())
            us = timeit(run)
            print(f"tokens {tokens:6d} dW {cout:5d}x{cin:5d} split {split:3d} tile {force}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s err {err:.1e}", flush=True)
