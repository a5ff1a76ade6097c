"""Timing of the attention core at the C2 step's spatial / text shapes (forward, backward; HIP events, back-to-back)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, t2v_amd
import t2v_amd.functional as F
bf = torch.bfloat16
def run(nb, heads, Sq, Sk, iters=20):
    C = heads * 64
    q = torch.randn(nb * Sq, C, device='cuda').to(bf).requires_grad_(); k = torch.randn(nb * Sk, C, device='cuda').to(bf).requires_grad_()
    v = torch.randn(nb * Sk, C, device='cuda').to(bf).requires_grad_(); do = torch.randn(nb * Sq, C, device='cuda').to(bf)
    ql, kl = F.SeqLayout(nb, Sq, Sq, 0, 1), F.SeqLayout(nb, Sk, Sk, 0, 1)
    fl = 4.0 * nb * heads * Sq * Sk * 64
    def t(fn):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / iters
    with torch.no_grad():
        tf = t(lambda: F.attention(q, k, v, heads, ql, kl))
    def fb():
        o = F.attention(q, k, v, heads, ql, kl); o.backward(do)
    tfb = t(fb)
    print(f"attn nb={nb} h={heads} Sq={Sq} Sk={Sk}: fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TF/s | fwd+bwd {tfb:8.1f} us {3.5 * fl / tfb / 1e6:7.1f} TF/s", flush=True)
for a in ((32, 5, 1024, 1024), (32, 10, 256, 256), (32, 20, 64, 64), (32, 5, 1024, 77), (32, 10, 256, 77), (24, 5, 2880, 2880), (4, 5, 9216, 9216)):
    run(*a)
