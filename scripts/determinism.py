import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, t2v_amd
import t2v_amd.functional as F
torch.manual_seed(0)
dev='cuda'
def rep(name, fn, n=4):
    outs=[fn() for _ in range(n)]
    torch.cuda.synchronize()
    base=outs[0]
    diffs=[max((a.float()-b.float()).abs().max().item() for a,b in zip(o,base)) for o in outs[1:]]
    print(f'{name:28s} max run-to-run abs diff {max(diffs):.3e}', flush=True)
bf=torch.bfloat16
# gemm dense
x=torch.randn(4096,640,device=dev).to(bf); w=torch.randn(640,640,device=dev)*0.04; b=torch.randn(640,device=dev); r=torch.randn(4096,640,device=dev).to(bf)
rep('gemm dense', lambda: (F.conv_linear(x,w,b,residual=r),))
# conv2d
xc=torch.randn(4*32*32,320,device=dev).to(bf); wc=torch.randn(320,320,3,3,device=dev)*0.02
cfg=F.ConvCfg.conv2d(4,32,32,3,1,1)
rep('conv2d 3x3', lambda: (F.conv_linear(xc,wc,None,cfg=cfg),))
# conv3d
cfg3=F.ConvCfg.conv3d_t(1,4,1024); w3=torch.randn(320,320,3,1,1,device=dev)*0.03
rep('conv3d t', lambda: (F.conv_linear(xc,w3,None,cfg=cfg3),))
# bwd of linear incl weight grad
def lin_bwd():
    xx=x.clone().requires_grad_(); ww=w.clone().requires_grad_()
    y=F.conv_linear(xx,ww,b); y.backward(r); return (xx.grad, ww.grad)
rep('linear bwd (dx, dw atomic)', lin_bwd)
def conv_bwd():
    xx=xc.clone().requires_grad_(); ww=wc.clone().requires_grad_()
    y=F.conv_linear(xx,ww,None,cfg=cfg); y.backward(torch.ones_like(y)); return (xx.grad, ww.grad)
rep('conv2d bwd', conv_bwd)
# groupnorm
gm=torch.randn(320,device=dev); bt=torch.randn(320,device=dev)
rep('groupnorm fwd', lambda: (F.group_norm(xc,gm,bt,32,1e-5,True,4),))
def gn_bwd():
    xx=xc.clone().requires_grad_(); y=F.group_norm(xx,gm,bt,32,1e-5,True,4); y.backward(xc); return (xx.grad,)
rep('groupnorm bwd', gn_bwd)
rep('layernorm fwd', lambda: (F.layer_norm(xc,gm,bt,1e-5),))
# attention spatial
q=torch.randn(4*1024,320,device=dev).to(bf); k=torch.randn(4*1024,320,device=dev).to(bf); v=torch.randn(4*1024,320,device=dev).to(bf)
lay=F.SeqLayout(4,1024,1024,0,1)
rep('attn spatial fwd', lambda: (F.attention(q,k,v,5,lay,lay),))
def attn_bwd(lq=lay, lk=lay, qq=q, kk=k, vv=v):
    a,b_,c=qq.clone().requires_grad_(),kk.clone().requires_grad_(),vv.clone().requires_grad_()
    o=F.attention(a,b_,c,5,lq,lk); o.backward(qq); return (a.grad,b_.grad,c.grad)
rep('attn spatial bwd', attn_bwd)
# temporal: B=1,F=4,HW=1024
layt=F.SeqLayout(1024,4,4*1024,1,1024,1024)
rep('attn temporal fwd', lambda: (F.attention(q,k,v,5,layt,layt),))
rep('attn temporal bwd', lambda: attn_bwd(layt,layt))
# cross
kt=torch.randn(77,320,device=dev).to(bf); vt=torch.randn(77,320,device=dev).to(bf)
layk=F.SeqLayout(4,77,77,0,1,4)
rep('attn cross fwd', lambda: (F.attention(q,kt,vt,5,lay,layk),))
rep('attn cross bwd', lambda: attn_bwd(lay,layk,q,kt,vt))
rep('geglu', lambda: (F.geglu(torch.cat([xc,xc],1)),))
# whole model twice
from test_train_gpu import _build
from oracle.weights import synthetic_batch
from t2v_amd.training import DenoiseTrainer
_, _, dunet, dvae, _ = _build(r=4)
batch = {k_: v_.cuda() for k_, v_ in synthetic_batch(4, 64, 64, seed=7, text_dim=64).items()}
tr = DenoiseTrainer(dunet, dvae, [p for p in dunet.parameters() if p.requires_grad], lr=1e-3)
for i in range(4):
    tr.opt.zero_grad(); l = tr._fwd_bwd(batch); torch.cuda.synchronize()
    print('eager run', i, 'loss', l.item(), 'gsum', tr.opt.flat_g.double().sum().item(), 'gabs', tr.opt.flat_g.double().abs().sum().item())
with torch.no_grad():
    lat=[]
    from t2v_amd.models.vae import tensor_to_vae_latent
    for i in range(3):
        lat.append(tensor_to_vae_latent(batch['pixel_values'], dvae, batch['vae_eps']))
    print('vae run-to-run', (lat[0]-lat[1]).abs().max().item(), (lat[0]-lat[2]).abs().max().item())
    outs=[dunet(batch['noise'], batch['timesteps'], batch['encoder_hidden_states']).sample for _ in range(3)]
    print('unet fwd run-to-run', (outs[0]-outs[1]).abs().max().item(), (outs[0]-outs[2]).abs().max().item(), 'out absmax', outs[0].abs().max().item())
