"""Per-shape cost of the dropped LoRA branch in its two forms (default train mode, utils/lora.py:57-62):
  epilogue form   fwd: lr2 launch (base + rank term in one)            bwd: dt kernel + lr1 launch
  pass form       fwd: ride launch (t as rank columns) + masked rank update of y      bwd: dt kernel + plain launch + rank update of dx
against the plain launch of the layer, median of 12 with the operands evicted from the L2s.
usage (GPU box): python scripts/branch_probe.py > gpurun_out/branch_probe.txt"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import t2v_amd  # noqa: E402,F401
import t2v_amd.functional as F  # noqa: E402
import t2v_amd.native as nv  # noqa: E402

BF = torch.bfloat16
big = None


def timeit(fn, reps=12):
    global big
    if big is None:
        big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(reps):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def problem(M, N, K, taps, g):
    cin = K // taps
    geom = None
    if taps == 9:
        geom = F.ConvCfg.conv2d(g[0], g[1], g[1], 3, 1, 1).fwd_geom(cin)
    elif taps == 3:
        geom = F.ConvCfg.conv3d_t(*g).fwd_geom(cin)
    a = torch.randn(M, cin, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    d = torch.empty(M, N, dtype=BF, device="cuda")
    b = torch.randn(N, device="cuda")
    kw = dict(M=M, N=N, K=K, A=a.data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, bias=b.data_ptr(),
              a_mode=1 if geom is not None else 0, geom=geom)
    return kw, [a, w, d, b]


# (name, M, N, K, taps, geometry): the wrapped Linear / Conv2d layers of the UNet at C2 (Conv3d wrappers never drop)
SHAPES = [
    ("L0 proj    ", 32768, 320, 320, 1, None),
    ("L0 qkv grp ", 32768, 960, 320, 1, None),
    ("L0 ff.proj ", 32768, 2560, 320, 1, None),
    ("L0 ff.out  ", 32768, 320, 1280, 1, None),
    ("L0 conv3x3 ", 32768, 320, 2880, 9, (32, 32)),
    ("L1 proj    ", 8192, 640, 640, 1, None),
    ("L1 qkv grp ", 8192, 1920, 640, 1, None),
    ("L1 ff.proj ", 8192, 5120, 640, 1, None),
    ("L1 ff.out  ", 8192, 640, 2560, 1, None),
    ("L1 conv3x3 ", 8192, 640, 5760, 9, (32, 16)),
    ("L2 proj    ", 2048, 1280, 1280, 1, None),
    ("L2 qkv grp ", 2048, 3840, 1280, 1, None),
    ("L2 ff.proj ", 2048, 10240, 1280, 1, None),
    ("L2 ff.out  ", 2048, 1280, 5120, 1, None),
    ("L2 conv3x3 ", 2048, 1280, 11520, 9, (32, 8)),
    ("L3 conv3x3 ", 512, 1280, 11520, 9, (32, 4)),
]
rp = 16
print("us: plain | fwd epilogue (lr2) | fwd pass form = ride + update | bwd-data epilogue (lr1) | bwd pass form = plain + update | dt kernel")
for name, M, N, K, taps, g in SHAPES:
    kw, keep = problem(M, N, K, taps, g)
    Dw = (torch.randn(rp, K, device="cuda") * K ** -0.5).to(BF)
    U = (torch.randn(rp, N, device="cuda") * 0.3).to(BF)            # up factor, GEMM layout of the pass form [rp, N]
    UT = U.t().contiguous()
    t = torch.empty(M, rp, dtype=BF, device="cuda")
    LA = torch.randn(M, rp, device="cuda").to(BF)
    LB = (torch.randn(N, taps * rp, device="cuda") * 0.3).to(BF)
    y = keep[2]
    plain = F.make_gemm(**kw)
    ride = F.make_gemm(**{**kw, "N": N + rp, "B2": Dw.data_ptr(), "ldb2": K, "n_split": N, "D2": t.data_ptr(), "ldd2": rp})
    s = nv.stream()
    tp = timeit(lambda: nv.call("t2v_gemm", C.byref(plain), s))
    tr = timeit(lambda: nv.call("t2v_gemm", C.byref(ride), s))
    try:
        lr2 = F.make_gemm(**kw, B2=Dw.data_ptr(), ldb2=K, D2=t.data_ptr(), ldd2=rp,
                          lr=dict(mode=2, rp=rp, b=UT.data_ptr(), ldb=rp, scale=1.0, drop_p=0.1, drop_seed=123))
        t2 = timeit(lambda: nv.call("t2v_gemm", C.byref(lr2), s))
    except RuntimeError:
        t2 = float("nan")
    try:
        lr1 = F.make_gemm(**kw, lr=dict(mode=1, rp=rp, taps=taps, a=LA.data_ptr(), lda=rp, b=LB.data_ptr(), ldb=taps * rp))
        t1 = timeit(lambda: nv.call("t2v_gemm", C.byref(lr1), s))
    except RuntimeError:
        t1 = float("nan")
    tu_drop = timeit(lambda: nv.call("t2v_lowrank_update_drop", y.data_ptr(), N, t.data_ptr(), rp, U.data_ptr(), N, M, N, rp, 1.0, 0.1, 123, s))
    tu = timeit(lambda: nv.call("t2v_lowrank_update", y.data_ptr(), N, t.data_ptr(), rp, U.data_ptr(), N, M, N, rp, 1.0, s)) if taps == 1 else float("nan")
    tdt = timeit(lambda: nv.call("t2v_lora_drop_dt", y.data_ptr(), N, U.data_ptr(), N, t.data_ptr(), rp, M, N, rp, 0.1, 123, s))
    gb = M * N * 2 / 1e3
    print(f"{name} M={M:6d} N={N:5d} K={K:6d} | plain {tp:6.1f} | lr2 {t2:6.1f} | ride {tr:6.1f} + upd {tu_drop:5.1f} = {tr + tu_drop:6.1f} | "
          f"lr1 {t1:6.1f} | plain + upd {tu:5.1f} = {tp + tu:6.1f} | dt {tdt:5.1f} ({gb / tdt / 1e3:4.2f} TB/s)", flush=True)
