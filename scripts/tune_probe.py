import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F, t2v_amd.native as nv
dev, bf = 'cuda', torch.bfloat16
def conv(nimg, H, W, Cin, Cout, t=False):
    cfg = F.ConvCfg.conv3d_t(2, nimg // 2, H * W) if t else F.ConvCfg.conv2d(nimg, H, W, 3, 1, 1)
    rows = nimg * H * W; taps = cfg.taps()
    a = torch.randn(rows, Cin, device=dev).to(bf); w = (torch.randn(Cout, taps * Cin, device=dev) * 0.02).to(bf)
    d = torch.empty(rows, Cout, device=dev, dtype=bf); g = cfg.fwd_geom(Cin)
    F.launch_gemm(M=rows, N=Cout, K=taps * Cin, A=a.data_ptr(), lda=Cin, B=w.data_ptr(), ldb=taps * Cin, D=d.data_ptr(), ldd=Cout, a_mode=1, geom=g)
def dense(M, N, K):
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf); d = torch.empty(M, N, device=dev, dtype=bf)
    F.launch_gemm(M=M, N=N, K=K, A=a.data_ptr(), lda=K, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N)
conv(32, 32, 32, 320, 320); conv(32, 16, 16, 640, 640); conv(32, 8, 8, 1280, 1280); conv(32, 4, 4, 1280, 1280)
conv(32, 32, 32, 320, 320, t=True); conv(32, 8, 8, 1280, 1280, t=True)
dense(32768, 336, 320); dense(32768, 2576, 320); dense(32768, 320, 1280); dense(8192, 656, 640); dense(2048, 1296, 1280); dense(2048, 10256, 1280); dense(2048, 1280, 5120)
conv(16, 256, 256, 128, 128); conv(16, 64, 64, 512, 512)
dense(8192, 8192, 8192)
torch.cuda.synchronize()
