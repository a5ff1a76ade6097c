"""Run ONE GEMM-family problem a few times (driver for rocprofv3 counter passes and for kernel-trace timing of a single
signature in isolation).   python scripts/gemm_shape_run.py M N K [taps] [rank_cols] [iters]
taps = 1 (dense), 3 ((3,1,1) temporal window over F=16), 9 (3x3 window); M = rows (for windows: 32 images of M/32 pixels)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
M, N, K = (int(a) for a in sys.argv[1:4])
taps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rc = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev, bf = 'cuda', torch.bfloat16
cin = K // taps
g = None
if taps == 9:
    side = int((M // 32) ** 0.5); cfg = F.ConvCfg.conv2d(32, side, side, 3, 1, 1); g = cfg.fwd_geom(cin)
elif taps == 3:
    cfg = F.ConvCfg.conv3d_t(2, 16, M // 32); g = cfg.fwd_geom(cin)
a = torch.randn(M, cin, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf)
d = torch.empty(M, N, device=dev, dtype=bf); b = torch.randn(N, device=dev)
kw = dict(M=M, N=N + rc, K=K, A=a.data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, bias=b.data_ptr(),
          a_mode=1 if g is not None else 0, geom=g)
if rc:
    w2 = (torch.randn(rc, K, device=dev) * 0.02).to(bf); t = torch.empty(M, rc, device=dev, dtype=bf)
    kw.update(B2=w2.data_ptr(), ldb2=K, n_split=N, D2=t.data_ptr(), ldd2=rc)
# a second buffer set, alternated, so that consecutive launches do not find their operands in the L2s
a2 = torch.randn(M, cin, device=dev).to(bf); d2 = torch.empty(M, N, device=dev, dtype=bf)
kw2 = dict(kw); kw2.update(A=a2.data_ptr(), D=d2.data_ptr())
for _ in range(3): F.launch_gemm(**kw); F.launch_gemm(**kw2)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(iters): F.launch_gemm(**(kw if i % 2 == 0 else kw2))
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / iters
print(f"M={M} N={N}+{rc} K={K} taps={taps}: {us:.1f} us/launch (back-to-back), {2.0 * M * (N + rc) * K / us / 1e6:.1f} TF/s", flush=True)
