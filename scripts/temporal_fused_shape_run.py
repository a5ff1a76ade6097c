"""Run ONE fused temporal unit a few times (driver for the counter passes of scripts/pmc_gemm_counters.sh).
    python scripts/temporal_fused_shape_run.py C batch frames hw [iters]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
from t2v_amd.models import leaves
C, B, Fr, hw = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
F._temporal_fused_maxc = 1 << 30
torch.manual_seed(0)
blk = leaves.BasicTransformerBlock(C, C // 64, 64, double_self_attention=True).cuda().eval()
for p in blk.parameters():
    p.requires_grad_(False)
t = (torch.randn(B * Fr * hw, C, device="cuda") * 1.2).to(torch.bfloat16)
qlay = F.SeqLayout(B * hw, Fr, Fr * hw, 1, hw, hw)
with torch.no_grad():
    for _ in range(iters):
        leaves._temporal_unit_fused(blk.norm1, blk.attn1, t, qlay)
torch.cuda.synchronize()
print("done")
