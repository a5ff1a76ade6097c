"""Where an in-step level-0 launch spends its time: the mode-2 forward launch of a dropped LoRA linear wrapper
(M x 320 x 320 + 16 ranks, bias, residual, mask, keep-bit plane) on the two configurations the step uses, timed in a graph
of ten launches and stamped per workgroup (T2V_W8_DBG=8: set-up / first stage / K loop / exchange + rank phase / output chunks /
store drain), next to the same launch without the mask, without the residual, and without the term (rank columns only).
    T2V_W8_DBG=8 python scripts/w8_epilogue_timeline.py [M [N [K]]]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2v_amd  # noqa: E402,F401
import t2v_amd.functional as F  # noqa: E402
import t2v_amd.native as nv  # noqa: E402

bf = torch.bfloat16
dev = "cuda"


def timeit(fn, n=10, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


def stamps(launch):
    ws = F._gemm_workspace()
    ws[16384: 16384 + 4096 * 16].zero_()
    launch()
    torch.cuda.synchronize()
    raw = ws[16384: 16384 + 4096 * 16].view(torch.int64).view(4096, 8).cpu().double()
    raw = raw[raw[:, 0] > 0]
    order = [0, 1, 2, 3, 5, 6, 7, 4]
    st = raw[:, order]
    dur = st[:, 1:] - st[:, :-1]
    names = ["setup(+hash)", "first stage", "K loop", "exchange", "slabs+rank phase", "output chunks", "store drain"]
    return f"{raw.shape[0]} wgs; cycles mean/max: " + "; ".join(f"{n} {float(dur[:, i].mean()):.0f}/{float(dur[:, i].max()):.0f}" for i, n in enumerate(names)) \
        + f"; whole {float((raw[:, 4] - raw[:, 0]).mean()):.0f}/{float((raw[:, 4] - raw[:, 0]).max()):.0f}"


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    K = int(sys.argv[3]) if len(sys.argv) > 3 else N
    rp = 16
    lib = nv.lib()
    g = torch.Generator().manual_seed(1)
    # two operand sets so that consecutive launches do not find their activations in the L2s
    sets = []
    for i in range(2):
        a = torch.randn(M, K, generator=g).to(bf).to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(bf).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(bf).to(dev)
        d = torch.empty(M, N, dtype=bf, device=dev)
        Dw = (torch.randn(rp, K, generator=g) * K ** -0.5).to(bf).to(dev)
        UT = (torch.randn(N, 16, generator=g) * 0.5).to(bf).to(dev)
        t = torch.zeros(M, rp, dtype=bf, device=dev)
        plane = torch.zeros(M * N // 16, dtype=torch.int16, device=dev)
        sets.append(dict(a=a, w=w, b=b, r=r, d=d, Dw=Dw, UT=UT, t=t, plane=plane))

    def desc(i, mask=True, res=True, term=True):
        s = sets[i]
        kw = dict(M=M, N=N, K=K, A=s["a"].data_ptr(), lda=K, B=s["w"].data_ptr(), ldb=K, D=s["d"].data_ptr(), ldd=N, bias=s["b"].data_ptr(),
                  a_mode=0, geom=None, R=s["r"].data_ptr() if res else None, ldr=N if res else 0)
        if term:
            lr = dict(mode=2, rp=rp, b=s["UT"].data_ptr(), ldb=16, scale=0.7, drop_p=0.1 if mask else 0.0, drop_seed=0x5EED)
            if mask:
                lr["plane"] = s["plane"].data_ptr()
            kw.update(B2=s["Dw"].data_ptr(), ldb2=K, D2=s["t"].data_ptr(), ldd2=rp, lr=lr)
        else:
            kw.update(B2=s["Dw"].data_ptr(), ldb2=K, n_split=N, D2=s["t"].data_ptr(), ldd2=rp, N=N + rp)
        return F.make_gemm(**kw)

    dbg = int(os.environ.get("T2V_W8_DBG", "0"))
    for name, opts in (("term+mask+res", dict()), ("term+mask", dict(res=False)), ("term, no mask, res", dict(mask=False)),
                       ("rank columns only, res", dict(term=False)), ("rank columns only", dict(term=False, res=False))):
        ds = [desc(i, **opts) for i in range(2)]
        for cfg in (14, 17):
            def launch(i, cfg=cfg):
                rc = lib.t2v_gemm_w8(C.byref(ds[i & 1]), cfg, 0, 1, nv.stream())
                assert rc == 0, lib.t2v_last_error().decode()
            us = timeit(launch)
            line = f"M={M} N={N} K={K} {name:28s} cfg {cfg}: {us:6.1f} us"
            if dbg & 8:
                line += " | " + stamps(lambda: launch(0))
            print(line, flush=True)


if __name__ == "__main__":
    main()
