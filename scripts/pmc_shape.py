"""Run ONE GEMM-family shape a few times (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: cheap, a handful of dispatches).
Default shape: the step's heaviest single layer type at C2 — ResnetBlock2D 3x3 conv at the 32x32 level, both UNet passes stacked
(rows = 2*16*32*32 = 32768, Cin = Cout = 320)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
nimg, H, W, Cin, Cout = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 32, 32, 320, 320)))
dev, bf = 'cuda', torch.bfloat16
cfg = F.ConvCfg.conv2d(nimg, H, W, 3, 1, 1)
rows = nimg * H * W
a = torch.randn(rows, Cin, device=dev).to(bf); w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(bf)
d = torch.empty(rows, Cout, device=dev, dtype=bf); b = torch.randn(Cout, device=dev)
g = cfg.fwd_geom(Cin)
def fn(): F.launch_gemm(M=rows, N=Cout, K=9 * Cin, A=a.data_ptr(), lda=Cin, B=w.data_ptr(), ldb=9 * Cin, D=d.data_ptr(), ldd=Cout, a_mode=1, geom=g, bias=b.data_ptr())
for _ in range(3): fn()          # includes the first-use autotune
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): fn()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
alg_bytes = (rows * Cin + rows * Cout + Cout * 9 * Cin) * 2
print(f"conv2d rows={rows} Cin={Cin} Cout={Cout}: {us:.1f} us/launch, {2.0*rows*Cout*9*Cin/us/1e6:.1f} TF/s, algorithmic bytes {alg_bytes/1e6:.1f} MB")
