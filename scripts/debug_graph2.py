import sys, os, copy; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, t2v_amd
from test_train_gpu import _build
from oracle.weights import synthetic_batch
from t2v_amd.training import DenoiseTrainer
_, _, dunet, dvae, _ = _build(r=4)
params = [p for p in dunet.parameters() if p.requires_grad]
names = [n for n, p in dunet.named_parameters() if p.requires_grad]
batch = {k: v.cuda() for k, v in synthetic_batch(4, 64, 64, seed=7, text_dim=64).items()}
t = DenoiseTrainer(dunet, dvae, params, lr=1e-3)
t.opt.zero_grad(); l_e = t._fwd_bwd(batch); torch.cuda.synchronize(); g_e = t.opt.flat_g.clone()
t.capture(batch, warmup=1)
t.opt.zero_grad(); t._graph.replay(); torch.cuda.synchronize(); l_g = t._static_loss.clone(); g_g = t.opt.flat_g.clone()
t.opt.zero_grad(); l_e2 = t._fwd_bwd(batch); torch.cuda.synchronize(); g_e2 = t.opt.flat_g.clone()
t.opt.zero_grad(); t._graph.replay(); torch.cuda.synchronize(); l_g2 = t._static_loss.clone(); g_g2 = t.opt.flat_g.clone()
print('loss eager', l_e.item(), l_e2.item(), 'graph', l_g.item(), l_g2.item())
print('grad eager-eager', (g_e-g_e2).abs().max().item(), 'graph-graph', (g_g-g_g2).abs().max().item(), 'eager-graph', (g_e-g_g).abs().max().item(), 'gnorm', g_e.norm().item(), g_g.norm().item())
off=0; bad=[]
for n,p in zip(names, params):
    k=p.numel(); d=(g_e[off:off+k]-g_g[off:off+k]).abs().max().item(); ref=g_e[off:off+k].abs().max().item()
    if d>1e-6*max(ref,1e-9)+1e-9: bad.append((d/(ref+1e-12), n, ref))
    off+=k
bad.sort(reverse=True)
print(len(bad), 'of', len(names), 'tensors differ; worst:', bad[:12])
