"""Yardstick: the step's top GEMM-family signatures on this library's kernels next to the vendor library (torch.matmul ->
hipBLASLt/rocBLAS) on the DENSE problem of the same M, N, K — for the windowed (conv) signatures the vendor library gets
the easier problem (no gather, no padding), so its number is an upper bound on what a library call could deliver there.
    python scripts/gemm_vs_library.py [iters]
Each line: ours us / TF/s | library us / TF/s.  Operands alternate between two buffer sets so that consecutive launches do
not find them in the L2s."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev, bf = "cuda", torch.bfloat16
# (M, N, rank columns, K, taps) — the heaviest signatures of a C2 step (T2V_BENCH_SHAPE_TABLE of bench.py)
SHAPES = [(32768, 320, 16, 320, 1), (32768, 320, 16, 960, 3), (32768, 320, 16, 2880, 9), (32768, 2560, 16, 320, 1),
          (32768, 320, 16, 2560, 1), (8192, 640, 16, 640, 1), (8192, 640, 16, 1920, 3), (8192, 640, 16, 5760, 9),
          (8192, 5120, 16, 640, 1), (2048, 1280, 16, 1280, 1), (2048, 1280, 16, 3840, 3), (2048, 1280, 16, 11520, 9),
          (2048, 10240, 16, 1280, 1), (2048, 1280, 16, 10240, 1), (512, 1280, 16, 3840, 3), (512, 1280, 16, 11520, 9),
          (2048, 1280, 16, 23040, 9)]


def timeit(fn):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for M, N, rc, K, taps in SHAPES:
    cin = K // taps
    g = None
    if taps == 9:
        side = int((M // 32) ** 0.5); g = F.ConvCfg.conv2d(32, side, side, 3, 1, 1).fwd_geom(cin)
    elif taps == 3:
        g = F.ConvCfg.conv3d_t(2, 16, M // 32).fwd_geom(cin)
    As = [torch.randn(M, cin, device=dev).to(bf) for _ in range(2)]
    Ds = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(2)]
    w = (torch.randn(N, K, device=dev) * 0.02).to(bf); w2 = (torch.randn(rc, K, device=dev) * 0.02).to(bf)
    t = torch.empty(M, rc, device=dev, dtype=bf); b = torch.randn(N, device=dev)
    kws = [dict(M=M, N=N + rc, K=K, A=As[i].data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=Ds[i].data_ptr(), ldd=N, bias=b.data_ptr(),
                a_mode=1 if g is not None else 0, geom=g, B2=w2.data_ptr(), ldb2=K, n_split=N, D2=t.data_ptr(), ldd2=rc) for i in range(2)]
    ours = timeit(lambda i: F.launch_gemm(**kws[i & 1]))
    Ad = [torch.randn(M, K, device=dev).to(bf) for _ in range(2)]
    wt = torch.cat([w, w2]).t().contiguous()           # (K, N+rc): the library's preferred NN operand
    wn = torch.cat([w, w2])                            # (N+rc, K): the layout the step holds (NT problem)
    outs = [torch.empty(M, N + rc, device=dev, dtype=bf) for _ in range(2)]
    lib_nn = timeit(lambda i: torch.matmul(Ad[i & 1], wt, out=outs[i & 1]))
    lib_nt = timeit(lambda i: torch.matmul(Ad[i & 1], wn.t(), out=outs[i & 1]))
    fl = 2.0 * M * (N + rc) * K
    lib = min(lib_nn, lib_nt)
    print(f"M={M:6d} N={N:5d}+{rc} K={K:5d} taps={taps}: ours {ours:7.1f} us {fl / ours / 1e6:6.1f} TF/s | library {lib:7.1f} us "
          f"{fl / lib / 1e6:6.1f} TF/s (nn {lib_nn:.1f}, nt {lib_nt:.1f}) | ours/library time {ours / lib:.2f}", flush=True)
    del As, Ds, Ad, outs
