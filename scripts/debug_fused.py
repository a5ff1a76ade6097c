import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, t2v_amd
from test_train_gpu import _build
from oracle.weights import synthetic_batch
from t2v_amd.training import DenoiseTrainer
_, _, dunet, dvae, _ = _build(r=4)
params = [p for p in dunet.parameters() if p.requires_grad]
names = [n for n, p in dunet.named_parameters() if p.requires_grad]
batch = {k: v.cuda() for k, v in synthetic_batch(4, 64, 64, seed=7, text_dim=64).items()}
t = DenoiseTrainer(dunet, dvae, params, lr=1e-3)
nb = sum(1 for m in dunet.modules() if hasattr(m, '_t2v_bank'))
t.opt.zero_grad(); l_f = t._fwd_bwd(batch); torch.cuda.synchronize(); g_f = t.opt.flat_g.clone()
ents = {}
for n, m in dunet.named_modules():
    if hasattr(m, '_t2v_bank'): ents[n] = m._t2v_bank; del m._t2v_bank
t.opt.zero_grad(); l_g = t._fwd_bwd(batch); torch.cuda.synchronize(); g_g = t.opt.flat_g.clone()
print('bank entries', nb, 'loss fused', l_f.item(), 'generic', l_g.item())
# per-param comparison via the param grads (views of flat_g)
off_err=[]
for n, p in zip(names, params):
    pass
import t2v_amd.lora_bank as lb
# compare in flat space
print('flat grad relerr', ((g_f-g_g).norm()/g_g.norm()).item(), 'norms', g_f.norm().item(), g_g.norm().item())
# per tensor: recompute views
t.opt.flat_g.copy_(g_f); gf = {n: p.grad.detach().clone() for n, p in zip(names, params)}
t.opt.flat_g.copy_(g_g); gg = {n: p.grad.detach().clone() for n, p in zip(names, params)}
bad = sorted(((float((gf[n]-gg[n]).norm()/(gg[n].norm()+1e-20)), n, float(gg[n].norm())) for n in names), reverse=True)
print('worst:', bad[:15])
kinds = {}
for e_, n, _ in bad:
    key = ('down' if 'lora_down' in n else 'up') + ('|' + ('conv3d' if 'temp_convs' in n else 'conv2d' if ('.conv' in n or 'conv_' in n or 'sampler' in n) else 'linear'))
    kinds.setdefault(key, []).append(e_)
for k, v in kinds.items(): print(k, 'n', len(v), 'max', max(v), 'median', sorted(v)[len(v)//2])
