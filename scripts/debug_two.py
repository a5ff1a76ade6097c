import sys, os, copy; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, t2v_amd
from test_train_gpu import _build
from oracle.weights import synthetic_batch
from t2v_amd.training import DenoiseTrainer
_, _, dunet, dvae, _ = _build(r=4)
batch = {k: v.cuda() for k, v in synthetic_batch(4, 64, 64, seed=7, text_dim=64).items()}
def fwd(m):
    with torch.no_grad():
        return m(batch['noise'], batch['timesteps'], batch['encoder_hidden_states']).sample
y0 = fwd(dunet)
dunet2 = copy.deepcopy(dunet)
y1 = fwd(dunet2); print('after deepcopy: dunet vs dunet2', (y0-y1).abs().max().item(), 'dunet again', (fwd(dunet)-y0).abs().max().item())
sd1, sd2 = dunet.state_dict(), dunet2.state_dict()
print('state dict equal', all(torch.equal(sd1[k], sd2[k]) for k in sd1))
p1 = [p for p in dunet.parameters() if p.requires_grad]; p2 = [p for p in dunet2.parameters() if p.requires_grad]
t1 = DenoiseTrainer(dunet, dvae, p1, lr=1e-3)
print('after t1: dunet', (fwd(dunet)-y0).abs().max().item(), 'dunet2', (fwd(dunet2)-y0).abs().max().item())
t2 = DenoiseTrainer(dunet2, dvae, p2, lr=1e-3)
print('after t2: dunet', (fwd(dunet)-y0).abs().max().item(), 'dunet2', (fwd(dunet2)-y0).abs().max().item())
sd1, sd2 = dunet.state_dict(), dunet2.state_dict()
print('state dict equal after rehome', all(torch.equal(sd1[k], sd2[k]) for k in sd1))
t1.opt.zero_grad(); la = t1._fwd_bwd(batch); print('t1 eager loss', la.item())
t2.opt.zero_grad(); lb = t2._fwd_bwd(batch); print('t2 eager loss', lb.item())
t2.capture(batch, warmup=1)
t1.opt.zero_grad(); la = t1._fwd_bwd(batch); print('t1 eager loss after t2 capture', la.item())
t2.opt.zero_grad(); t2._graph.replay(); print('t2 graph loss', t2._static_loss.item())
t1.opt.zero_grad(); la = t1._fwd_bwd(batch); print('t1 eager loss again', la.item())
