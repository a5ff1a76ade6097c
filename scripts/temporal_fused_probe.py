"""Time the one-launch temporal unit (csrc/temporal_fused.hip) against the separate launches of the training forward
(LayerNorm, fused q/k/v projection, FxF attention core, output projection + residual) at the temporal-attention signatures of a
config, ten calls per captured HIP graph, events around the replay, median of `reps`.  Prints one line per signature + totals over the units
of one UNet forward.   python scripts/temporal_fused_probe.py [c2|c4]"""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2v_amd  # noqa
import t2v_amd.functional as F
from t2v_amd.models import leaves

F._temporal_fused_maxc = 1 << 30          # time every width the library has a kernel for, not only those the policy dispatches
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
# (width, rows' pixel count per frame, units in one forward)  — ModelScope 1.7B: transformer_in (512 wide, 2 units), level 0 / 1 / 2
# with 5 temporal transformers each (2 down + 3 up), mid block 1; two units (attn1, attn2) per transformer
if cfg == "c2":
    Fr, B, grids = 16, 2, [(512, 32 * 32, 2), (320, 32 * 32, 10), (640, 16 * 16, 10), (1280, 8 * 8, 10), (1280, 4 * 4, 2)]   # B = 2: the two stacked passes / CFG pair
else:
    Fr, B, grids = 24, 2, [(512, 40 * 72, 2), (320, 40 * 72, 10), (640, 20 * 36, 10), (1280, 10 * 18, 10), (1280, 5 * 9, 2)]
reps = 30


def timed(fn, inner=10):
    """GPU time of one fn(): `inner` calls captured into ONE HIP graph (the host's launch path — ~40 us of Python per call here —
    stays out of the measurement, as it does in a captured sampling / training step), median over `reps` replays."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    for _ in range(3):
        g.replay()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2]


tot_f = tot_u = tot_fl = 0.0
for C, hw, units in grids:
    heads = C // 64
    torch.manual_seed(C + hw)
    blk = leaves.BasicTransformerBlock(C, heads, 64, double_self_attention=True).cuda().eval()
    for p in blk.parameters():
        p.requires_grad_(False)
    rows = B * Fr * hw
    t = (torch.randn(rows, C, device="cuda") * 1.2).to(torch.bfloat16)
    qlay = F.SeqLayout(B * hw, Fr, Fr * hw, 1, hw, hw)
    flops = rows * (8.0 * C * C + 4.0 * Fr * C)

    def unfused():
        n, r = F.layer_norm_res(t, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        return blk.attn1(n, qlay, residual=r)

    with torch.no_grad():
        tu = timed(unfused)
        ok = F.temporal_fused_ok(C, Fr, policy=False)
        tf = timed(lambda: leaves._temporal_unit_fused(blk.norm1, blk.attn1, t, qlay)) if ok else float("nan")
        if ok:
            a, b = unfused().float(), leaves._temporal_unit_fused(blk.norm1, blk.attn1, t, qlay).float()
            err = float((a - b).norm() / a.norm())
        else:
            err = float("nan")
    best = min(tu, tf) if ok else tu
    tot_u += units * tu; tot_f += units * best; tot_fl += units * flops
    print(f"{cfg} C={C:5d} rows={rows:7d} units={units:2d}: separate launches {tu:8.1f} us ({flops / tu * 1e-6:7.1f} TFLOP/s)   one launch {tf:8.1f} us "
          f"({flops / tf * 1e-6 if ok else float('nan'):7.1f} TFLOP/s = {flops / tf * 1e-6 / 2500 if ok else float('nan'):.3f} of bf16 MFMA peak)  fused-vs-separate relerr {err:.2e}", flush=True)
if os.environ.get("T2V_TF_ABLATE"):
    # where the time of one launch goes: the kernel with parts switched off (T2VTemporalFused.ablate)
    for C, hw, units in grids[:3]:
        if not F.temporal_fused_ok(C, Fr, policy=False):
            continue
        heads = C // 64
        blk = leaves.BasicTransformerBlock(C, heads, 64, double_self_attention=True).cuda().eval()
        for p in blk.parameters():
            p.requires_grad_(False)
        t = (torch.randn(B * Fr * hw, C, device="cuda") * 1.2).to(torch.bfloat16)
        qlay = F.SeqLayout(B * hw, Fr, Fr * hw, 1, hw, hw)
        row = []
        with torch.no_grad():
            for ab in (0, 1, 2, 3, 4, 8, 12, 13, 14, 15):
                F._temporal_ablate[0] = ab
                row.append(f"{ab}: {timed(lambda: leaves._temporal_unit_fused(blk.norm1, blk.attn1, t, qlay)):.1f}")
        F._temporal_ablate[0] = 0
        print(f"{cfg} C={C} rows={B * Fr * hw} ablations (bit0 no output pass, bit1 no LN input pass, bit2 no weight DMA, bit3 no MFMA loops) us: " + "  ".join(row), flush=True)
print(f"{cfg} all {sum(g[2] for g in grids)} units of one forward: separate {tot_u * 1e-3:.3f} ms ({tot_fl / tot_u * 1e-6:.1f} TFLOP/s), with the one-launch kernel where it "
      f"exists {tot_f * 1e-3:.3f} ms ({tot_fl / tot_f * 1e-6:.1f} TFLOP/s = {tot_fl / tot_f * 1e-6 / 2500:.3f} of peak)")
