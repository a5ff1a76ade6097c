"""Run ONE attention forward+backward problem a few times (driver for kernel-trace timing of the backward kernels in isolation).
python scripts/attn_bwd_shape_run.py nb heads Sq Sk [iters]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
nb, heads, Sq, Sk = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
C = heads * 64; bf = torch.bfloat16
mk = lambda rows: torch.randn(rows, C, device='cuda').to(bf).requires_grad_()
q, k, v = mk(nb * Sq), mk(nb * Sk), mk(nb * Sk); do = torch.randn(nb * Sq, C, device='cuda').to(bf)
ql, kl = F.SeqLayout(nb, Sq, Sq, 0, 1), F.SeqLayout(nb, Sk, Sk, 0, 1)
for _ in range(iters):
    F.attention(q, k, v, heads, ql, kl).backward(do)
torch.cuda.synchronize()
