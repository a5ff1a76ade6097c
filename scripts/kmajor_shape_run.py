"""Run ONE K-major (weight-gradient) GEMM a few times (driver for the counter passes of scripts/pmc_gemm_counters.sh).
    python scripts/kmajor_shape_run.py tokens cout cin split [iters]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
import t2v_amd.native as nv
tokens, cout, cin, split = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dy = torch.randn(tokens, cout, device="cuda").to(torch.bfloat16)
x = torch.randn(tokens, cin, device="cuda").to(torch.bfloat16)
dw = torch.zeros(cout, cin, dtype=torch.float32, device="cuda")
for _ in range(iters):
    F.launch_gemm(M=cout, N=cin, K=tokens, A=dy.data_ptr(), lda=cout, a_trans=1, B=x.data_ptr(), ldb=cin, b_trans=1,
                  D=dw.data_ptr(), ldd=cin, out_mode=nv.OUT_F32_ATOMIC, alpha=1.0, split_k=split)
torch.cuda.synchronize()
print("done")
