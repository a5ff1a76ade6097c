"""Per-shape timing of the GroupNorm / LayerNorm kernels at the C2 step's shapes (back-to-back launches, HIP events; two buffer
sets alternated so consecutive launches do not find their operands in the L2s)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F, t2v_amd.native as nv
bf = torch.bfloat16
def bench(name, fns, nbytes, iters=40):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"{name:46s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s", flush=True)
def gn(rows, C, nd, label):
    G = 32
    sets = []
    for _ in range(2):
        x = torch.randn(rows, C, device='cuda').to(bf); dy = torch.randn(rows, C, device='cuda').to(bf)
        y = torch.empty_like(x); dx = torch.empty_like(x); add = torch.randn(rows, C, device='cuda').to(bf)
        sums = torch.empty(nd * G * 2, device='cuda'); bs = torch.empty(nd * G * 2, device='cuda')
        sets.append((x, dy, y, dx, add, sums, bs))
    gm = torch.ones(C, device='cuda'); bt = torch.zeros(C, device='cuda')
    ws = F._gn_workspace(nd, G, torch.device('cuda', 0)); rpd = rows // nd; s = nv.stream()
    E = rows * C * 2
    mk = lambda f: [(lambda q=q: f(*q)) for q in sets]
    bench(f"gn_stats      {label}", mk(lambda x, dy, y, dx, add, sums, bs: nv.call("t2v_gn_stats", x.data_ptr(), C, nd, rpd, C, G, sums.data_ptr(), ws.data_ptr(), s)), E)
    bench(f"gn_apply      {label}", mk(lambda x, dy, y, dx, add, sums, bs: nv.call("t2v_gn_apply", x.data_ptr(), C, y.data_ptr(), C, nd, rpd, C, G, sums.data_ptr(), gm.data_ptr(), bt.data_ptr(), 1e-5, 1, 0.0, 0, s)), 2 * E)
    bench(f"gn_bwd_stats  {label}", mk(lambda x, dy, y, dx, add, sums, bs: nv.call("t2v_gn_bwd_stats", x.data_ptr(), C, dy.data_ptr(), C, nd, rpd, C, G, sums.data_ptr(), gm.data_ptr(), bt.data_ptr(), 1e-5, 1, 0.0, 0, bs.data_ptr(), ws.data_ptr(), None, None, None, s)), 2 * E)
    bench(f"gn_bwd_apply  {label}", mk(lambda x, dy, y, dx, add, sums, bs: nv.call("t2v_gn_bwd_apply", x.data_ptr(), C, dy.data_ptr(), C, dx.data_ptr(), C, nd, rpd, C, G, sums.data_ptr(), bs.data_ptr(), gm.data_ptr(), bt.data_ptr(), 1e-5, 1, 0.0, 0, add.data_ptr(), C, s)), 4 * E)
def ln(rows, C, label):
    sets = []
    for _ in range(2):
        x = torch.randn(rows, C, device='cuda').to(bf); dy = torch.randn(rows, C, device='cuda').to(bf)
        y = torch.empty_like(x); dx = torch.empty_like(x); add = torch.randn(rows, C, device='cuda').to(bf)
        st = torch.empty(rows * 2, device='cuda')
        sets.append((x, dy, y, dx, add, st))
    gm = torch.ones(C, device='cuda'); bt = torch.zeros(C, device='cuda'); s = nv.stream(); E = rows * C * 2
    mk = lambda f: [(lambda q=q: f(*q)) for q in sets]
    bench(f"ln_fwd        {label}", mk(lambda x, dy, y, dx, add, st: nv.call("t2v_layernorm_fwd", x.data_ptr(), C, y.data_ptr(), C, rows, C, gm.data_ptr(), bt.data_ptr(), 1e-5, st.data_ptr(), s)), 2 * E)
    bench(f"ln_bwd        {label}", mk(lambda x, dy, y, dx, add, st: nv.call("t2v_layernorm_bwd", x.data_ptr(), C, dy.data_ptr(), C, dx.data_ptr(), C, rows, C, gm.data_ptr(), st.data_ptr(), None, None, None, add.data_ptr(), C, s)), 4 * E)
for rows, C in ((32768, 320), (8192, 640), (2048, 1280), (512, 1280)):
    gn(rows, C, 32, f"per-frame  rows={rows} C={C}")
    gn(rows, C, 2, f"temporal   rows={rows} C={C}")
    ln(rows, C, f"rows={rows} C={C}")
