"""Driver of the K-loop cycle probe (profiles/r02_gemm_kloop_probe.txt): per-wave cycle split of the GEMM K loop into
wait+barrier / LDS-DMA issue / fragment reads + MFMA.  It needs a TEMPORARY instrumentation of csrc/gemm.hip that is not part of the
build: a `__device__ unsigned long long g_probe[8]`, clock64() stamps before the counted s_waitcnt, after the s_barrier, after
issue() and after compute() of `gemm_kernel_dma`'s K loop (lane 0 of every wave atomically adds its sums and iteration count), and
an exported `t2v_probe_read(unsigned long long* out, int reset)` that copies / clears the symbol."""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F, t2v_amd.native as nv
lib = nv.lib(); buf = (ctypes.c_ulonglong * 8)()
bf = torch.bfloat16
for (M, N, rc, K, taps, force) in [(32768, 320, 16, 2560, 1, "0,2,1"), (32768, 320, 16, 2880, 9, "0,2,1"), (2048, 1280, 16, 11520, 9, "2,2,1"), (8192, 640, 16, 5760, 9, "0,2,1"), (32768, 320, 16, 2560, 1, "0,0,1")]:
    os.environ["T2V_GEMM_FORCE_CFG"] = force
    cin = K // taps; g = None
    if taps == 9:
        side = int((M // 32) ** 0.5); g = F.ConvCfg.conv2d(32, side, side, 3, 1, 1).fwd_geom(cin)
    a = torch.randn(M, cin, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") * 0.02).to(bf); w2 = (torch.randn(rc, K, device="cuda") * 0.02).to(bf)
    d = torch.empty(M, N, device="cuda", dtype=bf); t = torch.empty(M, rc, device="cuda", dtype=bf)
    kw = dict(M=M, N=N + rc, K=K, A=a.data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, a_mode=1 if g is not None else 0, geom=g,
              B2=w2.data_ptr(), ldb2=K, n_split=N, D2=t.data_ptr(), ldd2=rc)
    for _ in range(3): F.launch_gemm(**kw)
    torch.cuda.synchronize(); lib.t2v_probe_read(buf, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): F.launch_gemm(**kw)
    e.record(); torch.cuda.synchronize(); lib.t2v_probe_read(buf, 1)
    wt, it, ct, nt, tot, nw = (buf[i] for i in range(6))
    print(f"M={M} N={N}+{rc} K={K} taps={taps} cfg={force}: {s.elapsed_time(e)*100:.1f} us/launch | per wave-iteration cycles: wait+barrier {wt/nt:.0f}, DMA issue {it/nt:.0f}, reads+MFMA {ct/nt:.0f} (16 MFMA = 512) | K loop {tot/nw:.0f} cycles/wave, {nt/nw:.1f} its", flush=True)
