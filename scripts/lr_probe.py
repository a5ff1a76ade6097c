"""Timings of the rank-wide epilogue term (T2VGemm.lr_mode) against the plain launches of the same shapes.
usage (GPU box): python scripts/lr_probe.py > gpurun_out/lr_probe.txt"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t2v_amd  # noqa: E402,F401
import t2v_amd.functional as F  # noqa: E402
import t2v_amd.native as nv  # noqa: E402

BF = torch.bfloat16


def timeit(fn, reps=20):
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(reps):
        big.zero_()                               # evict the operands from the L2s / MALL
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def problem(M, N, K, taps, nimg_hw=None):
    cin = K // taps
    geom = None
    if taps == 9:
        nimg, side = nimg_hw
        geom = F.ConvCfg.conv2d(nimg, side, side, 3, 1, 1).fwd_geom(cin)
    elif taps == 3:
        B, Fr, HW = nimg_hw
        geom = F.ConvCfg.conv3d_t(B, Fr, HW).fwd_geom(cin)
    a = torch.randn(M, cin, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    d = torch.empty(M, N, dtype=BF, device="cuda")
    b = torch.randn(N, device="cuda")
    kw = dict(M=M, N=N, K=K, A=a.data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, bias=b.data_ptr(),
              a_mode=1 if geom is not None else 0, geom=geom)
    return kw, [a, w, d, b]


SHAPES = [
    ("L0 proj   ", 32768, 320, 320, 1, None),
    ("L0 conv3x3", 32768, 320, 2880, 9, (32, 32)),
    ("L0 tconv  ", 32768, 320, 960, 3, (2, 16, 1024)),
    ("L0 ff.proj", 32768, 2560, 320, 1, None),
    ("L0 ff.out ", 32768, 320, 1280, 1, None),
    ("L1 proj   ", 8192, 640, 640, 1, None),
    ("L1 conv3x3", 8192, 640, 5760, 9, (32, 16)),
    ("L2 proj   ", 2048, 1280, 1280, 1, None),
    ("L2 conv3x3", 2048, 1280, 11520, 9, (32, 8)),
    ("L3 conv3x3", 512, 1280, 11520, 9, (32, 4)),
    ("L3 tconv  ", 512, 1280, 3840, 3, (2, 16, 16)),
]

rp, rk = 16, 16
for name, M, N, K, taps, g in SHAPES:
    kw, keep = problem(M, N, K, taps, g)
    Dw = (torch.randn(rp, K, device="cuda") * K ** -0.5).to(BF)
    UT = (torch.randn(N, rk, device="cuda") * 0.3).to(BF)
    t = torch.empty(M, rp, dtype=BF, device="cuda")
    LA = torch.randn(M, rp, device="cuda").to(BF)
    LB = (torch.randn(N, taps * rk, device="cuda") * 0.3).to(BF)
    plain = F.make_gemm(**kw)
    ride = F.make_gemm(**{**kw, "N": N + rp, "B2": Dw.data_ptr(), "ldb2": K, "n_split": N, "D2": t.data_ptr(), "ldd2": rp})
    lr2 = F.make_gemm(**kw, B2=Dw.data_ptr(), ldb2=K, D2=t.data_ptr(), ldd2=rp,
                      lr=dict(mode=2, rp=rp, b=UT.data_ptr(), ldb=rk, scale=1.0, drop_p=0.1, drop_seed=123))
    lr2n = F.make_gemm(**kw, B2=Dw.data_ptr(), ldb2=K, D2=t.data_ptr(), ldd2=rp,
                       lr=dict(mode=2, rp=rp, b=UT.data_ptr(), ldb=rk, scale=1.0, drop_p=0.0, drop_seed=0))
    lr1 = F.make_gemm(**kw, lr=dict(mode=1, rp=rp, taps=taps, a=LA.data_ptr(), lda=rp, b=LB.data_ptr(), ldb=taps * rk))
    s = nv.stream()
    out = [f"{name} M={M:6d} N={N:5d} K={K:6d}"]
    out.append(f"plain(table) {timeit(lambda: nv.call('t2v_gemm', C.byref(plain), s)):7.1f}")
    out.append(f"ride(table) {timeit(lambda: nv.call('t2v_gemm', C.byref(ride), s)):7.1f}")
    out.append(f"lr2(heur) {timeit(lambda: nv.call('t2v_gemm', C.byref(lr2), s)):7.1f}")
    out.append(f"lr2 nomask {timeit(lambda: nv.call('t2v_gemm', C.byref(lr2n), s)):7.1f}")
    out.append(f"lr1(heur) {timeit(lambda: nv.call('t2v_gemm', C.byref(lr1), s)):7.1f}")
    print(" | ".join(out), flush=True)
    # pinned configurations: plain vs lr2 on the same tile
    for cfg in (17, 22, 19, 20):
        row = [f"    cfg {cfg}:"]
        for nstep, sp in ((0, 1), (320, 1), (160, 1), (0, 2), (0, 4), (320, 4)):
            try:
                tp = timeit(lambda: nv.call("t2v_gemm_w8", C.byref(plain), cfg, nstep, sp, s), 8)
                t2 = timeit(lambda: nv.call("t2v_gemm_w8", C.byref(lr2), cfg, nstep, sp, s), 8)
                t1 = timeit(lambda: nv.call("t2v_gemm_w8", C.byref(lr1), cfg, nstep, sp, s), 8)
                row.append(f"[{nstep},{sp}] {tp:.0f}/{t2:.0f}/{t1:.0f}")
            except RuntimeError as e:
                row.append(f"[{nstep},{sp}] err")
        print(" ".join(row), flush=True)
