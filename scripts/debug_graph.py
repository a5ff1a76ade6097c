import sys, os, copy; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, t2v_amd
from test_train_gpu import _build
from oracle.weights import synthetic_batch
from t2v_amd.training import DenoiseTrainer
_, _, dunet, dvae, _ = _build(r=4)
dunet2 = copy.deepcopy(dunet)
p1 = [p for p in dunet.parameters() if p.requires_grad]; p2 = [p for p in dunet2.parameters() if p.requires_grad]
batch = {k: v.cuda() for k, v in synthetic_batch(4, 64, 64, seed=7, text_dim=64).items()}
t1 = DenoiseTrainer(dunet, dvae, p1, lr=1e-3); t2 = DenoiseTrainer(dunet2, dvae, p2, lr=1e-3)
t2.capture(batch, warmup=1)
for i in range(3):
    l1 = t1.train_step(batch); l2 = t2.replay_step(batch)
    torch.cuda.synchronize()
    print(i, 'eager', l1.item(), 'graph', l2.item(), 'gnorm', t1.opt.grad_norm().item(), t2.opt.grad_norm().item(),
          'pdiff', (t1.opt.flat_p - t2.opt.flat_p).abs().max().item(), 'gdiff', (t1.opt.flat_g - t2.opt.flat_g).abs().max().item(), 'steps', t1.opt.step_count.item(), t2.opt.step_count.item())
