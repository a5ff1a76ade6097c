"""Accurate per-shape timing of the GEMM family (HIP events over many back-to-back launches)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F, t2v_amd.native as nv
dev = 'cuda'
bf = torch.bfloat16
def bench(name, fn, flops, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    print(f'{name:58s} {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s', flush=True)
def dense(M, N, K, res=True, bias=True):
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    d = torch.empty(M, N, device=dev, dtype=bf); r = torch.randn(M, N, device=dev).to(bf) if res else None
    b = torch.randn(N, device=dev) if bias else None
    ws = F._gemm_workspace()
    def fn(): F.launch_gemm(M=M, N=N, K=K, A=a.data_ptr(), lda=K, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, bias=nv.ptr(b), R=nv.ptr(r), ldr=N)
    bench(f'dense M{M} N{N} K{K}', fn, 2.0 * M * N * K)
def conv(nimg, H, W, Cin, Cout, k=3, t=False):
    if t: cfg = F.ConvCfg.conv3d_t(1, nimg, H * W); rows = nimg * H * W
    else: cfg = F.ConvCfg.conv2d(nimg, H, W, k, 1, k // 2); rows = nimg * H * W
    taps = cfg.taps()
    a = torch.randn(rows, Cin, device=dev).to(bf); w = (torch.randn(Cout, taps * Cin, device=dev) * 0.02).to(bf)
    d = torch.empty(rows, Cout, device=dev, dtype=bf); b = torch.randn(Cout, device=dev)
    g = cfg.fwd_geom(Cin)
    def fn(): F.launch_gemm(M=rows, N=Cout, K=taps * Cin, A=a.data_ptr(), lda=Cin, B=w.data_ptr(), ldb=taps * Cin, D=d.data_ptr(), ldd=Cout, a_mode=1, geom=g, bias=b.data_ptr())
    bench(f'{"conv3d_t" if t else "conv2d"} rows{rows} Cin{Cin} Cout{Cout} taps{taps}', fn, 2.0 * rows * Cout * taps * Cin)
for M, C in ((16384, 320), (4096, 640), (1024, 1280), (256, 1280)):
    dense(M, C, C); dense(M, C + 16, C); dense(M, 8 * C, C); dense(M, C, 4 * C); dense(M, C, 16, bias=False); dense(M, 16, C, res=False, bias=False)
    conv(16, int((M // 16) ** 0.5), int((M // 16) ** 0.5), C, C); conv(16, int((M // 16) ** 0.5), int((M // 16) ** 0.5), C, C, t=True)
dense(1232, 320, 1024); dense(77, 1280, 1024)
conv(16, 32, 32, 640, 320); conv(16, 16, 16, 1920, 640); conv(16, 8, 8, 2560, 1280)
conv(16, 256, 256, 128, 128); conv(16, 128, 128, 256, 256); conv(16, 64, 64, 512, 512); conv(16, 32, 32, 512, 512)
dense(8192, 8192, 8192, res=False, bias=False); dense(4096, 4096, 4096, res=False, bias=False)
