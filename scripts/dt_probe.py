"""t2v_lora_drop_dt / t2v_lora_drop_dt_group per layer shape of a C2 step: us and effective TB/s (one read of dy), with the launch
shape the library picks and with the round-4 shape forced (T2V_DT_ROWG=2 + its WC rule) for comparison.
usage (GPU box): python scripts/dt_probe.py > gpurun_out/dt_probe.txt"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(32768, 320, 1), (32768, 320, 3), (32768, 1280, 1), (32768, 2560, 1), (8192, 640, 1), (8192, 640, 3), (8192, 2560, 1),
          (8192, 5120, 1), (2048, 1280, 1), (2048, 1280, 3), (2048, 5120, 1), (2048, 10240, 1), (512, 1280, 1), (512, 1280, 3),
          (154, 320, 2), (154, 1280, 2)]


def old_wc(M, N, nmem):
    nblk = (N + 63) // 64
    best, bw = 1e30, 1
    for wc in (1, 2, 4):
        blocks = (M + 32 * (4 // wc) - 1) // (32 * (4 // wc)) * nmem
        per = (nblk + wc - 1) // wc
        cost = per * wc / nblk * (1.0 if blocks >= 512 else 512.0 / blocks) * (1.05 if wc > 1 else 1.0)
        if cost < best:
            best, bw = cost, wc
    return bw


def child():
    import torch
    import t2v_amd  # noqa: F401
    import t2v_amd.native as nv
    BF = torch.bfloat16
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    M, N, nmem = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rp = 16
    dy = torch.randn(M, N * nmem, device="cuda").to(BF)
    U = (torch.randn(rp * nmem, N * nmem, device="cuda") * 0.3).to(BF)
    dt = torch.empty(M, rp * nmem, dtype=BF, device="cuda")
    seeds = (C.c_ulonglong * 3)(11, 22, 33)
    s = nv.stream()

    def run():
        if nmem == 1:
            nv.call("t2v_lora_drop_dt", dy.data_ptr(), N, U.data_ptr(), N, dt.data_ptr(), rp, M, N, rp, 0.1, 11, s)
        else:
            nv.call("t2v_lora_drop_dt_group", dy.data_ptr(), N * nmem, U.data_ptr(), N * nmem, rp * N * nmem + N, dt.data_ptr(), rp * nmem,
                    M, N, rp, nmem, 0.1, seeds, s)
    ts = []
    for _ in range(14):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{ts[len(ts) // 2]:.2f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    print("M N members | now us (TB/s) | round-4 launch shape us (TB/s)")
    for M, N, nmem in SHAPES:
        out = []
        for env in ({}, {"T2V_DT_ROWG": "2", "T2V_DT_WC": str(old_wc(M, N, nmem))}):
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(M), str(N), str(nmem)], env=e, capture_output=True, text=True)
            us = float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else float("nan")
            out.append(f"{us:7.2f} ({M * N * nmem * 2 / us / 1e6:4.2f})" if us == us else f"fail {r.stderr[-200:]}")
        print(f"{M:6d} {N:6d} {nmem} | {out[0]} | {out[1]}", flush=True)
