"""Probe of the 8-wave GEMM configurations (csrc/gemm_w8.hip) on the step's heavy signatures: every (configuration, column
step) candidate is checked against the table-selected kernel of t2v_gemm on the same operands and timed next to it
(back-to-back launches, operands alternating between two buffer sets so that no launch finds them in the L2s).
    python scripts/w8_probe.py [iters] [shape-filter]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import t2v_amd  # noqa: E402,F401
import t2v_amd.functional as F  # noqa: E402
import t2v_amd.native as nv  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev, bf = "cuda", torch.bfloat16
lib = nv.lib()

# (M, N, rank columns, K, taps, residual)
SHAPES = [
    (32768, 320, 16, 320, 1, 0), (32768, 320, 16, 960, 3, 0), (32768, 320, 16, 2880, 9, 0), (32768, 320, 16, 2560, 1, 0),
    (32768, 320, 16, 1280, 1, 1), (32768, 2560, 16, 320, 1, 0), (32768, 1280, 16, 320, 1, 0), (32768, 960, 48, 320, 1, 0),
    (8192, 640, 16, 640, 1, 0), (8192, 640, 16, 1920, 3, 0), (8192, 640, 16, 5760, 9, 0), (8192, 640, 16, 5120, 1, 0),
    (8192, 5120, 16, 640, 1, 0),
    (2048, 1280, 16, 1280, 1, 0), (2048, 1280, 16, 3840, 3, 0), (2048, 1280, 16, 11520, 9, 0), (2048, 1280, 16, 10240, 1, 0),
    (2048, 10240, 16, 1280, 1, 0), (2048, 1280, 48, 3840, 1, 0),
    (512, 1280, 16, 3840, 3, 0), (512, 1280, 16, 11520, 9, 0),
    (65536, 512, 0, 4608, 9, 1), (262144, 256, 0, 2304, 9, 1), (1048576, 128, 0, 1152, 9, 0),
]
BN = {0: 384, 1: 384, 2: 256, 3: 192, 4: 256, 5: 256, 6: 384, 7: 256, 8: 384, 9: 384, 10: 128, 11: 256, 12: 384, 13: 256, 14: 192, 15: 384, 16: 256, 17: 384, 18: 256, 19: 192, 20: 256, 21: 384, 22: 384}
BM = {0: 128, 1: 128, 2: 256, 3: 128, 4: 128, 5: 256, 6: 128, 7: 256, 8: 256, 9: 128, 10: 256, 11: 128, 12: 128, 13: 256, 14: 128, 15: 128, 16: 128, 17: 128, 18: 256, 19: 128, 20: 128, 21: 128, 22: 128}
ONLY = [int(x) for x in os.environ["W8_ONLY"].split(",")] if os.environ.get("W8_ONLY") else None


def candidates(M, Ntot, K):
    if os.environ.get("W8_CFGS"):                     # "cfg:step,cfg:step": pinned candidates (ablation runs)
        return [(int(x.split(":")[0]), int(x.split(":")[1]), 0, int((x.split(":") + ["1"])[2])) for x in os.environ["W8_CFGS"].split(",")]
    out = []
    for cfg in sorted(BN):
        if ONLY is not None and cfg not in ONLY:
            continue
        bn = BN[cfg]
        steps = {bn}
        for ntn in range(1, 64):                      # even splits of N into ntn tiles, rounded up to whole fragments
            st = ((Ntot + ntn - 1) // ntn + 31) // 32 * 32
            if st <= bn:
                steps.add(st)
                break
        for st in (160, 320):
            if st <= bn and Ntot > st:
                steps.add(st)
        for st in sorted(steps):
            ntn = 1
            while (ntn - 1) * st + bn < Ntot:
                ntn += 1
            wgs = -(-M // BM[cfg]) * ntn
            for sp in (1, 2, 3, 4, 6, 8):
                if sp > 1 and (wgs * sp > 320 or K // 64 // sp < 4):
                    continue
                if wgs * sp < 48:
                    continue
                out.append((cfg, st, wgs * sp, sp))
    return out


def timeit(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for M, N, rc, K, taps, res in SHAPES:
    tag = f"M={M} N={N}+{rc} K={K} taps={taps} res={res}"
    if flt and flt not in tag:
        continue
    cin = K // taps
    g = None
    if taps == 9:
        nimg = 32 if M < 65536 else 16
        side = int((M // nimg) ** 0.5)
        g = F.ConvCfg.conv2d(nimg, side, side, 3, 1, 1).fwd_geom(cin)
    elif taps == 3:
        g = F.ConvCfg.conv3d_t(2, 16, M // 32).fwd_geom(cin)
    torch.manual_seed(0)
    As = [torch.randn(M, cin, device=dev).to(bf) for _ in range(2)]
    Ds = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(2)]
    Rs = [torch.randn(M, N, device=dev).to(bf) for _ in range(2)] if res else [None, None]
    w = (torch.randn(N, K, device=dev) * 0.02).to(bf)
    b = torch.randn(N, device=dev)
    kws = []
    for i in range(2):
        kw = dict(M=M, N=N + rc, K=K, A=As[i].data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=Ds[i].data_ptr(), ldd=N, bias=b.data_ptr(),
                  a_mode=1 if g is not None else 0, geom=g, R=nv.ptr(Rs[i]), ldr=N if res else 0)
        if rc:
            w2 = (torch.randn(rc, K, device=dev) * 0.02).to(bf)
            t = torch.empty(M, rc, device=dev, dtype=bf)
            kw.update(B2=w2.data_ptr(), ldb2=K, n_split=N, D2=t.data_ptr(), ldd2=rc)
            kw["_keep"] = (w2, t)
        kws.append(kw)
    keep = [kw.pop("_keep", None) for kw in kws]
    descs = [F.make_gemm(**kw) for kw in kws]
    base = timeit(lambda i: nv.call("t2v_gemm", C.byref(descs[i & 1]), nv.stream()))
    fl = 2.0 * M * (N + rc) * K
    ref = Ds[0].float().clone()
    reft = keep[0][1].float().clone() if rc else None
    scale = float(ref.abs().max())
    print(f"{tag}: table kernel {base:7.1f} us {fl / base / 1e6:7.1f} TF/s", flush=True)
    best = (base, "table")
    for cfg, st, wgs, sp in candidates(M, N + rc, K):
        Ds[0].zero_()
        if rc:
            keep[0][1].zero_()
        rcode = lib.t2v_gemm_w8(C.byref(descs[0]), cfg, st, sp, nv.stream())
        if rcode != 0:
            print(f"    cfg {cfg} step {st}: rc={rcode} {lib.t2v_last_error().decode()}")
            continue
        torch.cuda.synchronize()
        err = float((Ds[0].float() - ref).abs().max()) / scale
        if rc:
            err = max(err, float((keep[0][1].float() - reft).abs().max()) / max(1e-9, float(reft.abs().max())))
        us = timeit(lambda i: lib.t2v_gemm_w8(C.byref(descs[i & 1]), cfg, st, sp, nv.stream()))
        if int(os.environ.get("T2V_W8_DBG", "0")) & 4:      # phase-cycle probe of the ping-pong schedule (per-wave totals)
            ws = F._gemm_workspace()
            nw = min(wgs if wgs else 256, 256) * 8
            raw = ws[: nw * 16].view(torch.int64).view(nw, 8).cpu().double()
            nph = raw[:, 5].clamp(min=1)
            per = (raw[:, :4] / nph[:, None])
            g0 = per[[i for i in range(nw) if (i % 8) < 4]].mean(0)
            g1 = per[[i for i in range(nw) if (i % 8) >= 4]].mean(0)
            print(f"      cycles per phase pair (M work, wait at M barrier, C issue, wait at C barrier): group0 "
                  f"{[round(float(x)) for x in g0]} sum {float(g0.sum()):.0f} | group1 {[round(float(x)) for x in g1]} sum {float(g1.sum()):.0f}")
        if int(os.environ.get("T2V_W8_DBG", "0")) & 8:      # timeline probe: per-workgroup stamps (cycles)
            ws = F._gemm_workspace()
            ws[16384: 16384 + 4096 * 16].zero_()
            lib.t2v_gemm_w8(C.byref(descs[0]), cfg, st, sp, nv.stream())
            torch.cuda.synchronize()
            raw = ws[16384: 16384 + 4096 * 16].view(torch.int64).view(4096, 8).cpu().double()
            raw = raw[raw[:, 0] > 0]
            # (the cycle counters of the 8 XCDs are not aligned with each other: phase DURATIONS per workgroup only)
            order = [0, 1, 2, 3, 5, 6, 7, 4]           # stamp slots in time order
            stamps = raw[:, order]
            dur = stamps[:, 1:] - stamps[:, :-1]
            names = ["setup", "first stage in flight", "K loop", "epilogue: bias/ticket + staging of pass 0", "pass-0 outputs issued",
                     "remaining passes", "wait for the stores"]
            print(f"      phases of {raw.shape[0]} workgroups (cycles; mean / max): "
                  + "; ".join(f"{n} {float(dur[:, i].mean()):.0f}/{float(dur[:, i].max()):.0f}" for i, n in enumerate(names))
                  + f"; whole workgroup {float((raw[:, 4] - raw[:, 0]).mean()):.0f}/{float((raw[:, 4] - raw[:, 0]).max()):.0f}")
        flag = "" if err < 2e-2 else "   <-- MISMATCH"
        print(f"    cfg {cfg} ({BM[cfg]}x{BN[cfg]}) step {st:3d} split {sp} wgs {wgs:4d}: {us:7.1f} us {fl / us / 1e6:7.1f} TF/s  x{base / us:4.2f}  err {err:.1e}{flag}",
              flush=True)
        if err < 2e-2 and us < best[0]:
            best = (us, f"cfg {cfg} step {st} split {sp}")
    print(f"  -> best {best[1]}: {best[0]:.1f} us {fl / best[0] / 1e6:.1f} TF/s (x{base / best[0]:.2f} vs table)", flush=True)
    del As, Ds, Rs
