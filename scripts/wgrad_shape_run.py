"""Run ONE t2v_lora_wgrad problem a few times (driver for counter passes / isolated timing).
   python scripts/wgrad_shape_run.py rows N C [taps] [iters]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C_
import torch, t2v_amd
import t2v_amd.functional as F, t2v_amd.native as nv
rows, N, Cc = (int(a) for a in sys.argv[1:4])
taps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
bf = torch.bfloat16
def mk():
    t = torch.randn(rows, 16, device='cuda').to(bf); dt = torch.randn(rows, 16, device='cuda').to(bf)
    dy = torch.randn(rows, N, device='cuda').to(bf); x = torch.randn(rows, Cc, device='cuda').to(bf)
    return t, dt, dy, x
sets = [mk(), mk()]
dU = torch.zeros(16, N, device='cuda'); dD = torch.zeros(16, taps * Cc, device='cuda')
g = None
if taps == 9:
    side = int((rows // 32) ** 0.5); g = F.ConvCfg.conv2d(32, side, side, 3, 1, 1).fwd_geom(Cc)
elif taps == 3:
    g = F.ConvCfg.conv3d_t(2, 16, rows // 32).fwd_geom(Cc)
def run(q):
    t, dt, dy, x = q
    w = nv.LoraWgrad()
    w.rows, w.rp, w.conv = rows, 16, 1 if g is not None else 0
    w.t, w.ldt, w.dy, w.lddy, w.N = t.data_ptr(), 16, dy.data_ptr(), N, N
    w.dU, w.lddu = dU.data_ptr(), N
    w.dt, w.lddt, w.x, w.ldx, w.C = dt.data_ptr(), 16, x.data_ptr(), Cc, Cc
    w.dD, w.lddd = dD.data_ptr(), taps * Cc
    if g is not None: w.geom = g
    w.alpha = 1.0
    nv.call("t2v_lora_wgrad", C_.byref(w), nv.stream())
for q in sets: run(q)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(iters): run(sets[i % 2])
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / iters
print(f"wgrad rows={rows} N={N} C={Cc} taps={taps}: {us:.1f} us/launch, {(rows * (N + Cc) * 2) / us / 1e6:.2f} TB/s algorithmic", flush=True)
