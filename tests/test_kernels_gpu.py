"""Kernel-level parity (GPU): every HIP op, forward and backward, against a plain PyTorch fp32 CPU reference
of the same op on the same bf16-rounded inputs.  Tolerance: bf16 storage => 2e-2 relative Frobenius error
on outputs and gradients (fp32 accumulate everywhere), stated per assert."""
import pytest
import torch
import torch.nn.functional as TF

from conftest import relerr

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _bf(t):
    return t.to(torch.bfloat16)


def _dev(t, rg=True):
    return t.detach().to("cuda").requires_grad_(rg)


def _cl(x4):  # NCHW -> token matrix
    n, c, h, w = x4.shape
    return x4.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def _uncl(m, n, h, w):
    return m.reshape(n, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (1000, 72, 136), (130, 8, 64), (4096, 640, 2560), (77, 1280, 1024),
                                   # M <= 96 with K % 64 == 0: the skinny weight-streaming kernel (gemm_skinny_kernel), forward and backward-data
                                   (2, 1280, 320), (33, 1000, 128), (96, 1024, 4096), (64, 72, 64), (77, 3072, 1024)])
def test_linear_fwd_bwd(M, N, K):
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)); w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g); res = _bf(torch.randn(M, N, generator=g)); dy = _bf(torch.randn(M, N, generator=g))
    xr = x.float().requires_grad_(); wr = _bf(w).float().requires_grad_(); br = b.clone().requires_grad_()
    rr = res.float().requires_grad_()
    yr = TF.linear(xr, wr, br) + rr
    yr.backward(dy.float())
    xd, wd, bd, rd = _dev(x), _dev(w), _dev(b), _dev(res)
    y = F.conv_linear(xd, wd, bd, residual=rd)
    y.backward(dy.cuda())
    assert relerr(y, yr) < TOL
    assert relerr(xd.grad, xr.grad) < TOL
    assert relerr(wd.grad, wr.grad) < TOL
    assert relerr(bd.grad, br.grad) < TOL
    assert relerr(rd.grad, rr.grad) < TOL


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad,up", [
    (2, 64, 72, 9, 11, 3, 1, 1, 0), (3, 32, 64, 8, 8, 3, 2, 1, 0), (2, 64, 64, 6, 5, 1, 1, 0, 0),
    (2, 32, 40, 5, 6, 3, 1, 1, 1), (1, 320, 320, 32, 32, 3, 1, 1, 0), (2, 16, 8, 7, 7, 3, 1, 1, 0)])
def test_conv2d_fwd_bwd(n, cin, cout, h, w, k, stride, pad, up):
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + h)
    x4 = _bf(torch.randn(n, cin, h, w, generator=g)); wt = torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5
    b = torch.randn(cout, generator=g)
    xr = x4.float().requires_grad_(); wr = _bf(wt).float().requires_grad_(); br = b.clone().requires_grad_()
    xin = TF.interpolate(xr, scale_factor=2.0, mode="nearest") if up else xr
    yr = TF.conv2d(xin, wr, br, stride=stride, padding=pad)
    dy4 = _bf(torch.randn(yr.shape, generator=g))
    yr.backward(dy4.float())
    cfg = F.ConvCfg.conv2d(n, h, w, k, stride, pad, up)
    assert (cfg.Ho, cfg.Wo) == tuple(yr.shape[2:])
    xd, wd, bd = _dev(_cl(x4)), _dev(wt), _dev(b)
    y = F.conv_linear(xd, wd, bd, cfg=cfg)
    y.backward(_cl(dy4).cuda().contiguous())
    assert relerr(_uncl(y, n, cfg.Ho, cfg.Wo), yr) < TOL
    assert relerr(_uncl(xd.grad, n, h, w), xr.grad) < TOL
    assert relerr(wd.grad, wr.grad) < TOL
    assert relerr(bd.grad, br.grad) < TOL


def test_conv2d_vae_pad01_stride2():
    """VAE Downsample2D(padding=0): F.pad(x,(0,1,0,1)) then 3x3 stride 2 pad 0."""
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(5)
    n, c, h, w = 2, 32, 8, 10
    x4 = _bf(torch.randn(n, c, h, w, generator=g)); wt = torch.randn(c, c, 3, 3, generator=g) * 0.06; b = torch.randn(c, generator=g)
    yr = TF.conv2d(TF.pad(x4.float(), (0, 1, 0, 1)), _bf(wt).float(), b, stride=2, padding=0)
    cfg = F.ConvCfg("conv", n, h, w, 3, 3, 2, 0, 0, 0, h // 2, w // 2)
    y = F.conv_linear(_dev(_cl(x4), False), _dev(wt, False), _dev(b, False), cfg=cfg)
    assert relerr(_uncl(y, n, h // 2, w // 2), yr) < TOL


@pytest.mark.parametrize("B,Fr,C,HW", [(1, 8, 64, 20), (2, 5, 32, 9), (1, 16, 320, 64)])
def test_conv3d_temporal(B, Fr, C, HW):
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(B + Fr + C)
    x5 = _bf(torch.randn(B, C, Fr, HW, 1, generator=g)); wt = torch.randn(C, C, 3, 1, 1, generator=g) * (3 * C) ** -0.5
    b = torch.randn(C, generator=g)
    xr = x5.float().requires_grad_(); wr = _bf(wt).float().requires_grad_()
    yr = TF.conv3d(xr, wr, b, padding=(1, 0, 0))
    dy5 = _bf(torch.randn(yr.shape, generator=g)); yr.backward(dy5.float())
    tok = lambda t: t[..., 0].permute(0, 2, 3, 1).reshape(B * Fr * HW, C).contiguous()   # (b,f,hw) rows
    untok = lambda m: m.reshape(B, Fr, HW, C).permute(0, 3, 1, 2).unsqueeze(-1)
    xd, wd = _dev(tok(x5)), _dev(wt)
    y = F.conv_linear(xd, wd, _dev(b, False), cfg=F.ConvCfg.conv3d_t(B, Fr, HW))
    y.backward(tok(dy5).cuda().contiguous())
    assert relerr(untok(y), yr) < TOL
    assert relerr(untok(xd.grad), xr.grad) < TOL
    assert relerr(wd.grad, wr.grad) < TOL


@pytest.mark.parametrize("nd,rows,C,G,silu", [(4, 63, 64, 32, True), (2, 1024, 320, 32, True), (1, 16 * 64, 640, 32, False),
                                             (3, 10, 128, 32, False), (2, 48, 2560, 32, True), (32, 1024, 320, 32, True)])
def test_groupnorm(nd, rows, C, G, silu):
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(nd + rows + C)
    x = _bf(torch.randn(nd * rows, C, generator=g) * 2 + 0.5); gm = torch.randn(C, generator=g); bt = torch.randn(C, generator=g)
    dy = _bf(torch.randn(nd * rows, C, generator=g))
    xr = x.float().requires_grad_(); gr = gm.clone().requires_grad_(); btr = bt.clone().requires_grad_()
    y3 = TF.group_norm(xr.view(nd, rows, C).permute(0, 2, 1), G, gr, btr, 1e-5)
    if silu:
        y3 = TF.silu(y3)
    yr = y3.permute(0, 2, 1).reshape(nd * rows, C)
    yr.backward(dy.float())
    xd, gd, bd = _dev(x), _dev(gm), _dev(bt)
    y = F.group_norm(xd, gd, bd, G, 1e-5, silu, nd)
    y.backward(dy.cuda())
    assert relerr(y, yr) < TOL
    assert relerr(xd.grad, xr.grad) < 3e-2
    assert relerr(gd.grad, gr.grad) < TOL
    assert relerr(bd.grad, btr.grad) < TOL


@pytest.mark.parametrize("rows,C", [(100, 64), (1024, 320), (257, 1280), (64, 512), (16384, 320)])
def test_layernorm(rows, C):
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(rows + C)
    x = _bf(torch.randn(rows, C, generator=g) + 0.3); gm = torch.randn(C, generator=g); bt = torch.randn(C, generator=g)
    dy = _bf(torch.randn(rows, C, generator=g))
    xr = x.float().requires_grad_(); gr = gm.clone().requires_grad_(); btr = bt.clone().requires_grad_()
    yr = TF.layer_norm(xr, (C,), gr, btr, 1e-5); yr.backward(dy.float())
    xd, gd, bd = _dev(x), _dev(gm), _dev(bt)
    y = F.layer_norm(xd, gd, bd, 1e-5); y.backward(dy.cuda())
    assert relerr(y, yr) < TOL
    assert relerr(xd.grad, xr.grad) < 3e-2
    assert relerr(gd.grad, gr.grad) < TOL
    assert relerr(bd.grad, btr.grad) < TOL


def _sdpa_ref(q, k, v, heads):
    # q: (nb, Sq, heads*64) fp32
    nb, Sq, _ = q.shape
    Sk = k.shape[1]
    qh = q.view(nb, Sq, heads, 64).transpose(1, 2); kh = k.view(nb, Sk, heads, 64).transpose(1, 2)
    vh = v.view(nb, Sk, heads, 64).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, -1)
    return (p @ vh).transpose(1, 2).reshape(nb, Sq, heads * 64)


# Sq >= 128 / Sk >= 128 take the workgroup kernels (4 waves sharing K|V resp. Q|dO tiles through LDS): whole tiles (256, 1024),
# ragged query and key blocks (200, 130), long queries against the 77 text keys, and the C4 / C5 spatial sequence lengths
# (40x72 = 2880, 72x128 = 9216)
@pytest.mark.parametrize("nb,heads,Sq,Sk", [(3, 2, 40, 40), (2, 5, 256, 256), (4, 1, 33, 77), (2, 3, 1024, 1024), (5, 2, 16, 16),
                                            (2, 2, 200, 130), (3, 1, 320, 77), (1, 2, 2880, 2880), (1, 1, 9216, 9216)])
def test_attention_spatial(nb, heads, Sq, Sk):
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(nb + heads + Sq + Sk)
    C = heads * 64
    q = _bf(torch.randn(nb, Sq, C, generator=g)); k = _bf(torch.randn(nb, Sk, C, generator=g)); v = _bf(torch.randn(nb, Sk, C, generator=g))
    do = _bf(torch.randn(nb, Sq, C, generator=g))
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    orf = _sdpa_ref(qr, kr, vr, heads); orf.backward(do.float())
    qd, kd, vd = _dev(q.view(-1, C)), _dev(k.view(-1, C)), _dev(v.view(-1, C))
    o = F.attention(qd, kd, vd, heads, F.SeqLayout(nb, Sq, Sq, 0, 1), F.SeqLayout(nb, Sk, Sk, 0, 1))
    o.backward(do.view(-1, C).cuda())
    assert relerr(o, orf.view(-1, C)) < TOL
    assert relerr(qd.grad, qr.grad.view(-1, C)) < 3e-2
    assert relerr(kd.grad, kr.grad.view(-1, C)) < 3e-2
    assert relerr(vd.grad, vr.grad.view(-1, C)) < 3e-2


@pytest.mark.parametrize("B,Fr,HW,heads", [(1, 16, 24, 2), (2, 8, 9, 5), (1, 24, 7, 1)])
def test_attention_temporal_strided(B, Fr, HW, heads):
    """Sequence over frames of a (b, f, hw) token matrix without any permute: stride = HW rows."""
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(B + Fr + HW)
    C = heads * 64
    q, k, v, do = (_bf(torch.randn(B, Fr, HW, C, generator=g)) for _ in range(4))
    tf = lambda t: t.permute(0, 2, 1, 3).reshape(B * HW, Fr, C)            # (b hw) f c
    tb = lambda t: t.view(B, HW, Fr, C).permute(0, 2, 1, 3)
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    orf = _sdpa_ref(tf(qr), tf(kr), tf(vr), heads); orf.backward(tf(do.float()))
    lay = F.SeqLayout(B * HW, Fr, Fr * HW, 1, HW, HW)
    qd, kd, vd = (_dev(t.reshape(-1, C)) for t in (q, k, v))
    o = F.attention(qd, kd, vd, heads, lay, lay)
    o.backward(do.reshape(-1, C).cuda())
    assert relerr(o.view(B, Fr, HW, C), tb(orf)) < TOL
    assert relerr(qd.grad.view(B, Fr, HW, C), qr.grad) < 3e-2
    assert relerr(kd.grad.view(B, Fr, HW, C), kr.grad) < 3e-2
    assert relerr(vd.grad.view(B, Fr, HW, C), vr.grad) < 3e-2


@pytest.mark.parametrize("B,Fr,HW,C", [(1, 16, 67, 320), (2, 24, 19, 640), (1, 8, 33, 512), (2, 5, 9, 64), (1, 4, 64, 128),
                                       (1, 32, 7, 320), (1, 16, 1024, 640)])
def test_temporal_unit_fused_forward(B, Fr, HW, C):
    """`x + attn(LN(x))` of a temporal BasicTransformerBlock (models/unet_3d_blocks.py:331-340 -> diffusers BasicTransformerBlock
    with double_self_attention) as ONE launch (csrc/temporal_fused.hip) against plain fp32 PyTorch on the same bf16-rounded
    operands; ragged tails (pixel counts that do not fill a block), clip lengths 4 .. 32 (24 = the Zeroscope clip: 8 idle row slots
    per block), every width the library has a kernel for."""
    import t2v_amd.functional as F
    assert F.temporal_fused_ok(C, Fr, policy=False)
    g = torch.Generator().manual_seed(B * 1000 + Fr * 10 + HW + C)
    heads = C // 64
    x = _bf(torch.randn(B, Fr, HW, C, generator=g) * 1.5 + 0.3)
    wq, wk, wv, wo = (_bf(torch.randn(C, C, generator=g) * C ** -0.5) for _ in range(4))
    bo = torch.randn(C, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr = x.float()
    n = TF.layer_norm(xr, (C,), gamma, beta, 1e-5)
    tf = lambda t: t.permute(0, 2, 1, 3).reshape(B * HW, Fr, C)            # (b hw) f c
    o = _sdpa_ref(tf(n @ wq.float().t()), tf(n @ wk.float().t()), tf(n @ wv.float().t()), heads)
    ref = xr + (o.view(B, HW, Fr, C).permute(0, 2, 1, 3) @ wo.float().t() + bo)
    out = F.temporal_attention_fused(x.reshape(-1, C).cuda(), gamma.cuda(), beta.cuda(), 1e-5, torch.cat([wq, wk, wv]).cuda(),
                                     F.temporal_fused_prepare_wo(wo.cuda()), bo.cuda(), B, Fr, HW)
    torch.cuda.synchronize()
    assert out.shape == (B * Fr * HW, C) and bool(torch.isfinite(out.float()).all())
    # the unit's own contribution (out - x) is what the kernel computes: bound it apart from the pass-through residual
    assert relerr(out.float().cpu().view(B, Fr, HW, C) - xr, ref - xr) < TOL
    assert relerr(out.view(B, Fr, HW, C), ref) < 1e-2


def test_temporal_block_no_grad_forward_takes_the_fused_unit_and_matches_the_training_forward():
    """A TransformerTemporalModel under torch.no_grad() (sampling: train.py:908-958) runs both attention units of its block through
    the one-launch kernel — with plain Linear projections and with cloneofsimo LoRA wrappers (W + s up down folded for the call) —
    and must agree with the SAME module's grad-enabled forward (separate LayerNorm / q,k,v / core / out-proj launches)."""
    import t2v_amd.functional as F
    from t2v_amd.models import leaves
    from t2v_amd.utils.lora import inject_trainable_lora_extended
    torch.manual_seed(5)
    m = leaves.TransformerTemporalModel(num_attention_heads=5, attention_head_dim=64, in_channels=320, num_layers=1).cuda()
    for p in m.parameters():
        p.requires_grad_(False)
    B, Fr, H, W = 1, 16, 6, 5
    x = leaves.Tok.from_nchw(torch.randn(B * Fr, 320, H, W).cuda())
    calls = []
    real = F.temporal_attention_fused
    F.temporal_attention_fused = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for lora in (False, True):
            if lora:
                inject_trainable_lora_extended(m, target_replace_module={"TransformerTemporalModel"}, r=8)
                for n_, p in m.named_parameters():
                    if "lora_up" in n_:
                        torch.nn.init.normal_(p, std=0.05)
                m.cuda().eval()                                  # (the wrappers' Dropout(0.1) off: sampling runs in eval mode)
            calls.clear()
            with torch.no_grad():
                y0 = m(x, num_frames=Fr).sample.m.float()
            assert len(calls) == 2, "both attention units of the block take the fused launch"
            y1 = m(x, num_frames=Fr).sample.m.float()           # grad mode: the training forward
            assert len(calls) == 2
            assert relerr(y0, y1) < 1e-2, (lora, relerr(y0, y1))
    finally:
        F.temporal_attention_fused = real


def test_attention_cross_shared_kv():
    """Text K/V shared by all frames of a video (encoder_hidden_states.repeat_interleave, unet_3d_condition.py:401)."""
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(3)
    B, Fr, S, heads, Sk = 2, 4, 48, 2, 77
    C = heads * 64
    q = _bf(torch.randn(B * Fr, S, C, generator=g)); k = _bf(torch.randn(B, Sk, C, generator=g)); v = _bf(torch.randn(B, Sk, C, generator=g))
    do = _bf(torch.randn(B * Fr, S, C, generator=g))
    qr, kr, vr = (t.float().requires_grad_() for t in (q, k, v))
    orf = _sdpa_ref(qr, kr.repeat_interleave(Fr, 0), vr.repeat_interleave(Fr, 0), heads); orf.backward(do.float())
    qd, kd, vd = _dev(q.view(-1, C)), _dev(k.view(-1, C)), _dev(v.view(-1, C))
    o = F.attention(qd, kd, vd, heads, F.SeqLayout(B * Fr, S, S, 0, 1), F.SeqLayout(B * Fr, Sk, Sk, 0, 1, Fr))
    o.backward(do.view(-1, C).cuda())
    assert relerr(o, orf.view(-1, C)) < TOL
    assert relerr(qd.grad, qr.grad.view(-1, C)) < 3e-2
    assert relerr(kd.grad, kr.grad.view(-1, C)) < 3e-2
    assert relerr(vd.grad, vr.grad.view(-1, C)) < 3e-2


def test_geglu_silu_concat():
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(300, 512, generator=g)); dy = _bf(torch.randn(300, 256, generator=g))
    xr = x.float().requires_grad_(); h, gate = xr.chunk(2, -1); yr = h * TF.gelu(gate); yr.backward(dy.float())
    xd = _dev(x); y = F.geglu(xd); y.backward(dy.cuda())
    assert relerr(y, yr) < TOL and relerr(xd.grad, xr.grad) < TOL
    xr2 = x.float().requires_grad_(); y2 = TF.silu(xr2); y2.backward(x.float())
    xd2 = _dev(x); ys = F.silu(xd2); ys.backward(x.cuda())
    assert relerr(ys, y2) < TOL and relerr(xd2.grad, xr2.grad) < TOL
    a = _bf(torch.randn(50, 64, generator=g)); b = _bf(torch.randn(50, 128, generator=g))
    ad, bd = _dev(a), _dev(b); c = F.concat(ad, bd); c.backward(torch.ones_like(c))
    assert torch.equal(c.cpu(), torch.cat([a, b], 1)) and ad.grad.shape == a.shape and bd.grad.shape == b.shape


def test_lora_composition_matches_reference_golden():
    """LoRA layers expressed with the native primitives reproduce the outputs of the REAL reference
    `utils/lora.py` layers (fixtures generated by tests/golden/make_golden.py)."""
    import os
    import t2v_amd.functional as F
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "lora_layers.pt"))
    for name in ("linear", "linear_nobias"):
        it = gold[name]; st = it["state"]
        x = _bf(it["x"]).reshape(-1, it["x"].shape[-1]).cuda()
        t = F.conv_linear(x, st["lora_down.weight"].cuda())
        y = F.conv_linear(x, st["linear.weight"].cuda(), st["linear.bias"].cuda() if "linear.bias" in st else None)
        y = F.conv_linear(t, st["lora_up.weight"].cuda(), residual=y, alpha=it["scale"])
        ref = it["y"].reshape(-1, it["y"].shape[-1])
        assert relerr(y[:, : ref.shape[1]], ref) < 3e-2, name


@pytest.mark.parametrize("M,N,r", [(300, 320, 16), (1024, 1288, 8), (77, 64, 32)])
def test_lowrank_update(M, N, r):
    """y += s * t @ U (the LoRA up-projection pass) against torch fp32."""
    import t2v_amd.native as nv
    g = torch.Generator().manual_seed(M + N + r)
    y = _bf(torch.randn(M, N, generator=g)); t = _bf(torch.randn(M, r, generator=g)); U = _bf(torch.randn(r, N, generator=g))
    ref = y.float() + 0.7 * (t.float() @ U.float())
    yd, td, Ud = y.cuda(), t.cuda(), U.cuda()
    nv.call("t2v_lowrank_update", yd.data_ptr(), N, td.data_ptr(), r, Ud.data_ptr(), N, M, N, r, 0.7, nv.stream())
    assert relerr(yd, ref) < 1e-2


@pytest.mark.parametrize("nimg,H,W,KH,KW,C,N,rp", [
    (1, 1, 1, 1, 1, 320, 320, 16),        # linear (rows set below), C and N multiples of 64
    (1, 1, 1, 1, 1, 72, 1288, 8),         # linear, ragged column tiles, rank 8
    (3, 8, 8, 3, 3, 64, 128, 16),         # conv 3x3 pad 1
    (2, 5, 7, 3, 3, 24, 40, 32),          # conv 3x3, odd image, two rank passes
    (2, 4, 48, 3, 1, 128, 128, 16),       # (3,1,1) Conv3d as a 3x1 window over (B, F, H*W)
    (1, 16, 16, 3, 3, 8, 320, 24),        # conv_in-like (Cin padded to 8), ranks 16 + 8
])
def test_lora_wgrad(nimg, H, W, KH, KW, C, N, rp):
    """t2v_lora_wgrad: dU = a t^T dy and dD[j,tap,c] = a sum dt[p(q,tap), j] x[q, c] against torch fp32 (autograd's
    weight gradients of lora_up / lora_down, utils/lora.py:57-62,134-139,211-216).  Outputs accumulate (+=)."""
    import ctypes as C_
    import t2v_amd.native as nv
    conv = KH * KW > 1
    rows = nimg * H * W if conv else 1000
    g = torch.Generator().manual_seed(rows + C + N + rp)
    t = _bf(torch.randn(rows, rp, generator=g)); dy = _bf(torch.randn(rows, N, generator=g))
    dt = _bf(torch.randn(rows, rp, generator=g)); x = _bf(torch.randn(rows, C, generator=g))
    taps = KH * KW
    dU0 = torch.randn(rp, N, generator=g); dD0 = torch.randn(rp, taps * C, generator=g)
    a = 0.5
    refU = dU0 + a * (t.float().T @ dy.float())
    if conv:
        py, px = KH // 2, KW // 2
        xi = x.float().view(nimg, H, W, C)
        xp = torch.zeros(nimg, H + KH - 1, W + KW - 1, C); xp[:, py:py + H, px:px + W] = xi
        di = dt.float().view(nimg, H, W, rp)
        parts = [torch.einsum('nhwj,nhwc->jc', di, xp[:, ky:ky + H, kx:kx + W]) for ky in range(KH) for kx in range(KW)]
        refD = dD0 + a * torch.stack(parts, 1).reshape(rp, taps * C)
    else:
        refD = dD0 + a * (dt.float().T @ x.float())
    td, dyd, dtd, xd, dU, dD = t.cuda(), dy.cuda(), dt.cuda(), x.cuda(), dU0.cuda(), dD0.cuda()
    w = nv.LoraWgrad()
    w.rows, w.rp, w.conv = rows, rp, int(conv)
    w.t, w.ldt, w.dy, w.lddy, w.N = td.data_ptr(), rp, dyd.data_ptr(), N, N
    w.dU, w.lddu = dU.data_ptr(), N
    w.dt, w.lddt, w.x, w.ldx, w.C = dtd.data_ptr(), rp, xd.data_ptr(), C, C
    w.dD, w.lddd = dD.data_ptr(), taps * C
    if conv:
        w.geom = nv.ConvGeom(C, H, W, H, W, KH, KW, 1, 1, KH // 2, KW // 2, 1, 0)
    w.alpha = a
    nv.call("t2v_lora_wgrad", C_.byref(w), nv.stream())
    torch.cuda.synchronize()
    eu, ed = relerr(dU, refU), relerr(dD, refD)
    print('lora_wgrad relerr', eu, ed)
    assert eu < 1e-3 and ed < 1e-3        # fp32 accumulation of exact bf16 products: only summation order differs


@pytest.mark.parametrize("M,N,rp,ldu_extra", [(1000, 320, 16, 0), (4096, 640, 16, 640), (77, 1280, 8, 0), (513, 2560, 32, 0),
                                               (32, 72, 24, 0), (8192, 320, 16, 0)])
def test_lora_drop_dt_and_masked_wgrad(M, N, rp, ldu_extra):
    """Backward of a dropped LoRA branch without a masked copy of dy (utils/lora.py:49,57-62 autograd): t2v_lora_drop_dt gives
    dt = (mask . dy / (1-p)) U from ONE masked pass over dy, and T2VLoraWgrad.drop_p makes the dU contraction regenerate the
    same mask (oracle/dropout.keep_mask, index = row * N + column).  fp32 references on the bf16 operands."""
    import ctypes as C_
    import t2v_amd.native as nv
    from oracle.dropout import keep_mask
    p, seed = 0.1, 0xABCDEF123 + M
    g = torch.Generator().manual_seed(M + N + rp)
    dy = _bf(torch.randn(M, N, generator=g)); U = _bf(torch.randn(rp, N + ldu_extra, generator=g) * 0.3)
    t = _bf(torch.randn(M, rp, generator=g)); x = _bf(torch.randn(M, 64, generator=g)); dt_in = _bf(torch.randn(M, rp, generator=g))
    m = keep_mask(seed, M, N, p).float() / (1.0 - p)
    gm = dy.float() * m
    ref_dt = gm @ U[:, :N].float().T
    dyd, Ud, dt = dy.cuda(), U.cuda(), torch.full((M, rp), 7.0, dtype=torch.bfloat16, device="cuda")
    nv.call("t2v_lora_drop_dt", dyd.data_ptr(), N, Ud.data_ptr(), N + ldu_extra, dt.data_ptr(), rp, M, N, rp, p, seed, nv.stream())
    torch.cuda.synchronize()
    e = relerr(dt, ref_dt)
    print("lora_drop_dt relerr", e)
    assert e < 6e-3                       # exact bf16 products, fp32 sums, one bf16 rounding of the result
    # masked dU (and an unmasked dD in the same descriptor)
    a = 0.5
    dU0 = torch.randn(rp, N, generator=g); dD0 = torch.randn(rp, 64, generator=g)
    refU = dU0 + a * (t.float().T @ gm)
    refD = dD0 + a * (dt_in.float().T @ x.float())
    td, dtd, xd, dU, dD = t.cuda(), dt_in.cuda(), x.cuda(), dU0.cuda(), dD0.cuda()
    w = nv.LoraWgrad()
    w.rows, w.rp, w.conv = M, rp, 0
    w.t, w.ldt, w.dy, w.lddy, w.N = td.data_ptr(), rp, dyd.data_ptr(), N, N
    w.dU, w.lddu = dU.data_ptr(), N
    w.dt, w.lddt, w.x, w.ldx, w.C = dtd.data_ptr(), rp, xd.data_ptr(), 64, 64
    w.dD, w.lddd = dD.data_ptr(), 64
    w.alpha, w.drop_p, w.drop_seed = a, p, seed
    nv.call("t2v_lora_wgrad", C_.byref(w), nv.stream())
    torch.cuda.synchronize()
    eu, ed = relerr(dU, refU), relerr(dD, refD)
    print("masked wgrad relerr", eu, ed)
    assert eu < 1e-3 and ed < 1e-3


def test_lora_wgrad_batch_equals_per_layer_launches():
    """t2v_lora_wgrad_batch: the descriptors of several layers (linear, 3x3, (3,1,1); one or two rank passes; different row
    counts) in ONE launch give what the per-layer launches give (fp32 atomics: summation order only)."""
    import ctypes as C_
    import t2v_amd.native as nv
    g = torch.Generator().manual_seed(2024)
    layers = [(1, 1, 1000, 1, 1, 64, 128, 16), (3, 8, 8, 3, 3, 64, 128, 16), (2, 4, 48, 3, 1, 128, 128, 16), (1, 1, 4096, 1, 1, 320, 320, 16),
              (2, 5, 7, 3, 3, 24, 40, 32), (1, 1, 77, 1, 1, 1024, 320, 16), (1, 16, 16, 3, 3, 8, 320, 24)]
    descs, outs, keep = [], [], []
    for nimg, H, W, KH, KW, C, N, rp in layers:
        conv = KH * KW > 1
        rows = nimg * H * W if conv else W
        taps = KH * KW
        t = _bf(torch.randn(rows, rp, generator=g)).cuda(); dy = _bf(torch.randn(rows, N, generator=g)).cuda()
        dt = _bf(torch.randn(rows, rp, generator=g)).cuda(); x = _bf(torch.randn(rows, C, generator=g)).cuda()
        pair = []
        for _ in range(2):                   # [0]: per-layer launches, [1]: the batch
            dU = torch.zeros(rp, N, device="cuda"); dD = torch.zeros(rp, taps * C, device="cuda")
            w = nv.LoraWgrad()
            w.rows, w.rp, w.conv = rows, rp, int(conv)
            w.t, w.ldt, w.dy, w.lddy, w.N = t.data_ptr(), rp, dy.data_ptr(), N, N
            w.dU, w.lddu = dU.data_ptr(), N
            w.dt, w.lddt, w.x, w.ldx, w.C = dt.data_ptr(), rp, x.data_ptr(), C, C
            w.dD, w.lddd = dD.data_ptr(), taps * C
            if conv:
                w.geom = nv.ConvGeom(C, H, W, H, W, KH, KW, 1, 1, KH // 2, KW // 2, 1, 0)
            w.alpha = 0.7
            pair.append((w, dU, dD))
        keep.append((t, dy, dt, x))
        nv.call("t2v_lora_wgrad", C_.byref(pair[0][0]), nv.stream())
        descs.append(pair[1][0]); outs.append((pair[0][1], pair[0][2], pair[1][1], pair[1][2]))
    n = len(descs)
    arr = (nv.LoraWgrad * n)(*descs)
    nbytes = int(nv.lib().t2v_lora_wgrad_batch_bytes(n))
    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True); dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    nv.call("t2v_lora_wgrad_batch", arr, n, host.data_ptr(), dev.data_ptr(), nbytes, nv.stream())
    torch.cuda.synchronize()
    for dU1, dD1, dU2, dD2 in outs:
        assert float(dU1.abs().sum()) > 0 and float(dD1.abs().sum()) > 0
        assert relerr(dU2, dU1) < 1e-5 and relerr(dD2, dD1) < 1e-5


def test_norm_passthrough_residual_gradient():
    """group_norm_res / layer_norm_res: the second output is x for its residual use; dx = norm_bwd(dy) + d(residual)
    is formed inside the backward kernel (t2v_gn_bwd_apply / t2v_layernorm_bwd `addend`)."""
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(11)
    nd, rows, C, G = 2, 96, 320, 32
    x = _bf(torch.randn(nd * rows, C, generator=g) + 0.3); gm = torch.randn(C, generator=g); bt = torch.randn(C, generator=g)
    dy = _bf(torch.randn(nd * rows, C, generator=g)); dr = _bf(torch.randn(nd * rows, C, generator=g))
    # GroupNorm(+SiLU)
    xr = x.float().requires_grad_()
    y3 = TF.silu(TF.group_norm(xr.view(nd, rows, C).permute(0, 2, 1), G, gm, bt, 1e-5)).permute(0, 2, 1).reshape(nd * rows, C)
    (y3 * dy.float()).sum().backward(retain_graph=True); (xr * dr.float()).sum().backward()
    xd = _dev(x)
    y, xres = F.group_norm_res(xd, gm.cuda(), bt.cuda(), G, 1e-5, True, nd)
    assert xres.data_ptr() == xd.data_ptr()
    torch.autograd.backward([y, xres], [dy.cuda(), dr.cuda()])
    assert relerr(y, y3) < TOL and relerr(xd.grad, xr.grad) < 3e-2
    xd2 = _dev(x)                              # pass-through output unused: plain backward
    y2, _ = F.group_norm_res(xd2, gm.cuda(), bt.cuda(), G, 1e-5, True, nd)
    y2.backward(dy.cuda())
    xr2 = x.float().requires_grad_()
    y4 = TF.silu(TF.group_norm(xr2.view(nd, rows, C).permute(0, 2, 1), G, gm, bt, 1e-5)).permute(0, 2, 1).reshape(nd * rows, C)
    y4.backward(dy.float())
    assert relerr(xd2.grad, xr2.grad) < 3e-2
    # LayerNorm
    xr = x.float().requires_grad_()
    yl = TF.layer_norm(xr, (C,), gm, bt, 1e-5)
    (yl * dy.float()).sum().backward(retain_graph=True); (xr * dr.float()).sum().backward()
    xd = _dev(x)
    y, xres = F.layer_norm_res(xd, gm.cuda(), bt.cuda(), 1e-5)
    torch.autograd.backward([y, xres], [dy.cuda(), dr.cuda()])
    assert relerr(y, yl) < TOL and relerr(xd.grad, xr.grad) < 3e-2


@pytest.mark.parametrize("nimg,H,W,KH,KW,C,r", [(2, 8, 8, 3, 3, 64, 16), (3, 5, 7, 3, 3, 40, 8), (2, 4, 48, 3, 1, 128, 16),
                                               (1, 6, 6, 3, 3, 24, 32)])
def test_lowrank_window_update(nimg, H, W, KH, KW, C, r):
    """y[q,c] += s sum_tap sum_j t[p(q,tap), j] D[j, tap, c]  (backward-data of a LoRA down conv) against torch fp32."""
    import ctypes as C_
    import t2v_amd.native as nv
    rows, taps = nimg * H * W, KH * KW
    g = torch.Generator().manual_seed(rows + C + r)
    y = _bf(torch.randn(rows, C, generator=g)); t = _bf(torch.randn(rows, r, generator=g))
    D = _bf(torch.randn(r, taps, C, generator=g) * 0.3)
    py, px = KH // 2, KW // 2
    ti = t.float().view(nimg, H, W, r)
    tp = torch.zeros(nimg, H + KH - 1, W + KW - 1, r); tp[:, py:py + H, px:px + W] = ti
    ref = y.float().clone().view(nimg, H, W, C)
    for ky in range(KH):
        for kx in range(KW):
            # p = q - (ky - py, kx - px)  ->  padded index (iy - ky + 2py, ix - kx + 2px)
            sh = tp[:, 2 * py - ky: 2 * py - ky + H, 2 * px - kx: 2 * px - kx + W]
            ref += 0.5 * sh @ D.float()[:, ky * KW + kx, :]
    yd, td, Dd = y.cuda(), t.cuda(), D.cuda()
    geom = nv.ConvGeom(C, H, W, H, W, KH, KW, 1, 1, py, px, 1, 0)
    nv.call("t2v_lowrank_window_update", yd.data_ptr(), C, td.data_ptr(), r, Dd.data_ptr(), taps * C, C_.byref(geom), rows, C, r,
            0.5, nv.stream())
    assert relerr(yd, ref.view(rows, C)) < 1e-2


def test_dropout_mask_protocol_matches_cpu_restatement():
    """The counter-based keep decision (csrc/common.h drop_keep) is bit-identical to oracle/dropout.py — the mask that is
    'identically defined on CPU and GPU' (SURVEY 8d)."""
    import t2v_amd.native as nv
    from oracle.dropout import keep_mask
    rows, cols, p, seed = 257, 320, 0.3, 0x1234ABCD5
    x = torch.ones(rows, cols, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    nv.call("t2v_dropout_mask", x.data_ptr(), cols, y.data_ptr(), cols, rows, cols, p, seed, nv.stream())
    keep = keep_mask(seed, rows, cols, p)
    assert torch.equal(y.cpu() != 0, keep)
    assert torch.allclose(y.cpu().float()[keep], torch.tensor(1.0 / (1.0 - p)).to(torch.bfloat16).float())


def test_dropout_epoch_offsets_every_dropout_kernel_like_the_restatement():
    """t2v_set_dropout_epoch: launches issued while a device counter is registered use seed ^ splitmix64(counter + G) — the
    counter is read when the kernel RUNS (a captured graph sees its current value) — for the mask kernel, the GroupNorm epilogue
    and the GEMM epilogue; unregistering restores the plain protocol."""
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    from oracle.dropout import effective_seed, keep_mask
    rows, cols, p, seed = 96, 320, 0.3, 0xABCDEF12345
    ctr = torch.tensor([7], dtype=torch.int64, device="cuda")
    x = torch.ones(rows, cols, dtype=torch.bfloat16, device="cuda")
    try:
        nv.call("t2v_set_dropout_epoch", ctr.data_ptr())
        y7 = torch.empty_like(x)
        nv.call("t2v_dropout_mask", x.data_ptr(), cols, y7.data_ptr(), cols, rows, cols, p, seed, nv.stream())
        g = torch.cuda.CUDAGraph()                               # the captured launch carries the ADDRESS: replays follow the counter
        yg = torch.empty_like(x)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            nv.call("t2v_dropout_mask", x.data_ptr(), cols, yg.data_ptr(), cols, rows, cols, p, seed, nv.stream())
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            nv.call("t2v_dropout_mask", x.data_ptr(), cols, yg.data_ptr(), cols, rows, cols, p, seed, nv.stream())
        masks = []
        for e in (8, 9):
            ctr.fill_(e)
            g.replay()
            torch.cuda.synchronize()
            masks.append((yg.cpu() != 0).clone())
        gm = torch.randn(cols); bt = torch.randn(cols)          # GroupNorm(+SiLU)+dropout epilogue under epoch 9
        xin = _bf(torch.randn(2 * 48, cols))
        yn = F.group_norm(xin.cuda(), gm.cuda(), bt.cuda(), 32, 1e-5, True, 2, p, seed)
        td = _bf(torch.randn(rows, 16)); w = _bf(torch.randn(cols, 16) * 0.3)
        yl = F.conv_linear(td.cuda(), w.cuda(), None, F.LINEAR, None, torch.zeros(rows, cols, dtype=torch.bfloat16, device="cuda"),
                           alpha=1.0, drop_p=p, drop_seed=seed)
    finally:
        nv.call("t2v_set_dropout_epoch", None)
    y0 = torch.empty_like(x)
    nv.call("t2v_dropout_mask", x.data_ptr(), cols, y0.data_ptr(), cols, rows, cols, p, seed, nv.stream())
    assert torch.equal(y7.cpu() != 0, keep_mask(effective_seed(seed, 7), rows, cols, p))
    assert torch.equal(masks[0], keep_mask(effective_seed(seed, 8), rows, cols, p))
    assert torch.equal(masks[1], keep_mask(effective_seed(seed, 9), rows, cols, p)) and not torch.equal(masks[0], masks[1])
    assert torch.equal(y0.cpu() != 0, keep_mask(seed, rows, cols, p))
    k9 = keep_mask(effective_seed(seed, 9), rows, cols, p)
    ref = TF.silu(TF.group_norm(xin.float().view(2, 48, cols).permute(0, 2, 1), 32, gm, bt, 1e-5)).permute(0, 2, 1).reshape(96, cols)
    nz = ref.abs() > 1e-3                                       # (a SiLU output that rounds to zero says nothing about the mask)
    assert torch.equal((yn.detach().cpu() != 0)[nz], k9[nz])
    prod = td.float() @ w.float().t()
    nz = prod.abs() > 1e-3
    assert torch.equal((yl.detach().cpu() != 0)[nz], k9[nz])


def test_groupnorm_dropout_matches_masked_reference():
    """GroupNorm+SiLU+Dropout (TemporalConvLayer conv2-4, models/unet_3d_blocks.py:312…) with the protocol mask, fwd + bwd."""
    import t2v_amd.functional as F
    from oracle.dropout import keep_mask
    nd, rows, C, G, p, seed = 2, 96, 320, 32, 0.1, 777
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(nd * rows, C, generator=g) + 0.2); gm = torch.randn(C, generator=g); bt = torch.randn(C, generator=g)
    dy = _bf(torch.randn(nd * rows, C, generator=g))
    m = keep_mask(seed, nd * rows, C, p).float() / (1.0 - p)
    xr = x.float().requires_grad_()
    y3 = TF.silu(TF.group_norm(xr.view(nd, rows, C).permute(0, 2, 1), G, gm, bt, 1e-5)).permute(0, 2, 1).reshape(nd * rows, C) * m
    y3.backward(dy.float())
    xd = _dev(x)
    y = F.group_norm(xd, gm.cuda(), bt.cuda(), G, 1e-5, True, nd, p, seed)
    y.backward(dy.cuda())
    assert torch.equal(y.detach().cpu() != 0, (y3 != 0))          # same elements dropped
    assert relerr(y, y3) < TOL and relerr(xd.grad, xr.grad) < 3e-2


@pytest.mark.parametrize("flavour", ["linear", "conv2d", "conv3d"])
def test_stable_lora_layers_against_the_reference_formulas(flavour):
    """a22: the stable_lora layer flavours on the native path against the reference's forward FORMULAS written out here with
    torch.nn.functional (not through the product's own module mirror):
      Conv2d  (stable_lora/lora.py:119-126): conv2d(x, W + (B@A).view(W.shape) * alpha/r)
      Conv3d  (:141-149,190-197): conv3d(x, W + mean((B@A).view(out, in, k, k, 1), dim=-2, keepdim) * alpha/r), kernel (k,1,1)
      Linear  (loralib, :199-207): x W^T + (x A^T B^T) * alpha/r
    forward, input gradient and both factor gradients."""
    import torch.nn.functional as TF_
    from t2v_amd.models import leaves
    from t2v_amd.stable_lora import lora as SL
    import t2v_amd.functional as F
    torch.manual_seed(11)
    r, alpha = 4, 8.0
    if flavour == "linear":
        mod = SL.Linear(64, 96, r=r, lora_alpha=alpha).train()
        x5 = torch.randn(300, 64)
    elif flavour == "conv2d":
        mod = SL.Conv2d(32, 48, 3, r=r, lora_alpha=alpha, padding=1).train()
        x5 = torch.randn(2, 32, 6, 10)
    else:
        mod = SL.Conv3d(32, 32, 3, r=r, lora_alpha=alpha, padding=(1, 0, 0)).train()
        x5 = torch.randn(2, 32, 5, 4, 6)                         # (B, C, F, H, W)
    with torch.no_grad():
        mod.lora_B.copy_(torch.randn(mod.lora_B.shape) * 0.1)
    x5 = _bf(x5)
    W, b = _bf(mod.weight.detach()).float(), mod.bias.detach().float()
    A, B = mod.lora_A.detach().clone().requires_grad_(), mod.lora_B.detach().clone().requires_grad_()
    xr = x5.float().requires_grad_()
    sc = alpha / r
    if flavour == "linear":
        yr = xr @ W.t() + b + (xr @ A.t() @ B.t()) * sc
        xt, cfg, back = x5, F.LINEAR, (lambda m: m)
        tok = lambda t: t
    elif flavour == "conv2d":
        yr = TF_.conv2d(xr, W + (B @ A).view(W.shape) * sc, b, padding=1)
        n, c, h, w = x5.shape
        cfg = F.ConvCfg.conv2d(n, h, w, 3, 1, 1)
        tok = lambda t: t.permute(0, 2, 3, 1).reshape(n * h * w, -1)
    else:
        o, i, k = W.shape[:3]
        yr = TF_.conv3d(xr, W + torch.mean((B @ A).view(o, i, k, k, 1), dim=-2, keepdim=True) * sc, b, padding=(1, 0, 0))
        bsz, c, f, h, w = x5.shape
        cfg = F.ConvCfg.conv3d_t(bsz, f, h * w)
        tok = lambda t: t.permute(0, 2, 3, 4, 1).reshape(bsz * f * h * w, -1)
    dy = _bf(torch.randn(yr.shape))
    yr.backward(dy.float())
    dev = mod.cuda()
    xd = _dev(tok(x5))
    y = leaves.run_layer(dev, xd, cfg)
    y.backward(tok(dy).cuda().contiguous())
    assert relerr(y, tok(yr.detach())) < TOL
    assert relerr(xd.grad, tok(xr.grad)) < 3e-2
    assert relerr(dev.lora_A.grad, A.grad) < 3e-2 and relerr(dev.lora_B.grad, B.grad) < 3e-2


def test_loralib_linear_input_dropout_on_the_lowrank_branch():
    """loralib.Linear (stable_lora's Linear flavour, stable_lora/lora.py:199-207): y = x W^T + (lora_dropout(x) A^T B^T) * alpha/r.
    The mask sits on the branch INPUT; forward and the three gradients against torch fp32 with the protocol mask."""
    from oracle.dropout import keep_mask
    from t2v_amd.models import leaves
    from t2v_amd.stable_lora.lora import Linear
    torch.manual_seed(5)
    M, Cin, Cout, r, p = 200, 64, 96, 8, 0.25
    lin = Linear(Cin, Cout, r=r, lora_alpha=16, lora_dropout=p).train()
    with torch.no_grad():
        lin.lora_B.copy_(torch.randn(Cout, r) * 0.2)
    x = _bf(torch.randn(M, Cin)); dy = _bf(torch.randn(M, Cout))
    leaves.set_dropout_seed(77)
    seed = (77 * 1000003 + 1) & 0xFFFFFFFFFFFF
    m = keep_mask(seed, M, Cin, p).float() / (1.0 - p)
    xr = x.float().requires_grad_()
    A, B = lin.lora_A.detach().clone().requires_grad_(), lin.lora_B.detach().clone().requires_grad_()
    yr = xr @ _bf(lin.weight.detach()).float().t() + lin.bias.detach() + ((xr * m) @ A.t() @ B.t()) * lin.scaling
    yr.backward(dy.float())
    dev = lin.cuda()
    xd = _dev(x)
    y = leaves.run_layer(dev, xd)
    y.backward(dy.cuda())
    assert relerr(y, yr) < TOL and relerr(xd.grad, xr.grad) < 3e-2
    assert relerr(dev.lora_A.grad, A.grad) < 3e-2 and relerr(dev.lora_B.grad, B.grad) < 3e-2
    dev.eval()                                   # eval: dropout off (and loralib merges the delta into the weight)
    y2 = leaves.run_layer(dev, _dev(x))
    assert relerr(y2, x.float() @ (lin.weight.detach().float().cpu()).t() + lin.bias.detach().cpu()) < TOL


def test_lora_branch_dropout_in_gemm_epilogue():
    """y = residual + dropout(scale * up(t)) (utils/lora.py:57-62 with dropout_p > 0): mask applied in the GEMM epilogue and
    re-applied to dy in backward."""
    import t2v_amd.functional as F
    from oracle.dropout import keep_mask
    M, N, r, p, seed, s = 300, 320, 16, 0.1, 4242, 0.7
    g = torch.Generator().manual_seed(8)
    t = _bf(torch.randn(M, r, generator=g)); w = _bf(torch.randn(N, r, generator=g) * 0.3); res = _bf(torch.randn(M, N, generator=g))
    dy = _bf(torch.randn(M, N, generator=g))
    m = keep_mask(seed, M, N, p).float() / (1.0 - p)
    tr, wr = t.float().requires_grad_(), w.float().requires_grad_()
    yr = res.float() + (s * (tr @ wr.t())) * m
    yr.backward(dy.float())
    td, wd = _dev(t), _dev(w)
    y = F.conv_linear(td, wd, None, F.LINEAR, None, res.cuda(), alpha=s, drop_p=p, drop_seed=seed)
    y.backward(dy.cuda())
    assert relerr(y, yr) < TOL
    assert relerr(td.grad, tr.grad) < 3e-2 and relerr(wd.grad, wr.grad) < 3e-2


@pytest.mark.parametrize("n,max_norm,world", [(100003, 1.0, 1), (4096, 0.05, 2), (777, 0.0, 1)])
def test_sumsq_adamw_match_torch_optim(n, max_norm, world):
    """t2v_sumsq + t2v_adamw against `clip_grad_norm_` + `torch.optim.AdamW` on IDENTICAL gradients (train.py:868-877:
    global-norm clip 1.0, AdamW betas (0.9,0.999), wd 1e-2, eps 1e-8), several steps, incl. the 1/world pre-scale of a
    SUM all-reduce and the no-clip case."""
    import t2v_amd.native as nv
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-2
    pr = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([pr], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    p = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda"); ss = torch.zeros(1, device="cuda")
    ws = torch.zeros(2048, device="cuda")
    for it in range(4):
        grad = torch.randn(n, generator=g) * (10.0 if it == 1 else 0.01)       # one step far above the clip threshold
        pr.grad = (grad / world).clone()
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_([pr], max_norm)
        opt.step()
        gd = grad.cuda()
        ss.zero_()
        if max_norm > 0:
            nv.call("t2v_sumsq", gd.data_ptr(), n, ss.data_ptr(), ws.data_ptr(), nv.stream())
            ss2 = torch.zeros(1, device="cuda")
            nv.call("t2v_sumsq", gd.data_ptr(), n, ss2.data_ptr(), ws.data_ptr(), nv.stream())
            assert torch.equal(ss, ss2)                                  # fixed-order reduction: bit-reproducible
            # the kernel clips the SCALED gradient: sumsq is taken after the all-reduce, before the 1/world scale
            assert abs(ss.sqrt().item() / world - total.item()) < 1e-4 * total.item()
        nv.call("t2v_adamw", p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, wd,
                ss.data_ptr() if max_norm > 0 else None, max_norm, 1.0 / world, step.data_ptr(), nv.stream())
        torch.cuda.synchronize()
        assert step.item() == it + 1
        upd_ref = (pr.detach() - p0).double()
        upd = (p.cpu() - p0).double()
        assert float((upd - upd_ref).norm() / upd_ref.norm()) < 1e-4, it
    assert relerr(p, pr) < 1e-6


@pytest.mark.parametrize("fused", [False, True])
def test_lora_wrappers_all_kinds_match_reference_golden(fused):
    """All six layer kinds of tests/golden/lora_layers.pt (outputs of the REAL reference `utils/lora.py` Linear / Conv2d 3x3,
    stride-2, 1x1 / Conv3d (3,1,1) wrappers) through the PRODUCT path: the drop-in wrapper modules evaluated by
    `models.leaves.run_layer` — unfused (three launches) and fused (trainer bank attached: the path a train step takes)."""
    import os
    from t2v_amd.functional import ConvCfg, LINEAR
    from t2v_amd.models.leaves import Tok, run_layer
    from t2v_amd.training import FlatAdamW
    from t2v_amd.utils import lora as L
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "lora_layers.pt"))
    for name, it in gold.items():
        st, x, r = it["state"], it["x"], it["r"]
        if name.startswith("linear"):
            w = st["linear.weight"]
            mod = L.LoraInjectedLinear(w.shape[1], w.shape[0], bias="linear.bias" in st, r=r, dropout_p=0.0)
        elif name == "conv3d":
            w = st["conv.weight"]
            mod = L.LoraInjectedConv3d(w.shape[1], w.shape[0], (3, 1, 1), (1, 0, 0), r=r, dropout_p=0.0)
        else:
            w = st["conv.weight"]
            k = w.shape[2]
            mod = L.LoraInjectedConv2d(w.shape[1], w.shape[0], k, 2 if name == "conv2d_s2" else 1, 1 if k == 3 else 0, r=r, dropout_p=0.0)
        mod.load_state_dict(st, strict=True)
        mod.scale = it["scale"]
        mod = mod.cuda().train()
        base = mod.linear if name.startswith("linear") else mod.conv
        base.requires_grad_(False)
        holder = torch.nn.Module(); holder.layer = mod
        if fused:
            FlatAdamW([mod.lora_down.weight, mod.lora_up.weight], model=holder)
            assert getattr(mod, "_t2v_bank", None) is not None
        xq = _bf(x)
        if name.startswith("linear"):
            xm, cfg = xq.reshape(-1, xq.shape[-1]).cuda(), LINEAR
            ref = it["y"].reshape(-1, it["y"].shape[-1])
        elif name == "conv3d":
            B, C, Fr, H, W = xq.shape
            xm = xq.permute(0, 2, 3, 4, 1).reshape(B * Fr * H * W, C).contiguous().cuda()
            cfg = ConvCfg.conv3d_t(B, Fr, H * W)
            ref = it["y"].permute(0, 2, 3, 4, 1).reshape(B * Fr * H * W, -1)
        else:
            n, C, H, W = xq.shape
            xm = _cl(xq).cuda()
            cfg = ConvCfg.conv2d(n, H, W, w.shape[2], 2 if name == "conv2d_s2" else 1, 1 if w.shape[2] == 3 else 0)
            ref = _cl(it["y"])
        xm.requires_grad_(True)
        y = run_layer(mod, xm, cfg)
        e = relerr(y[:, : ref.shape[1]].float(), ref)
        assert e < 3e-2, (name, fused, e)
        if fused:       # backward through the fused node: dx against autograd of the reference expression in fp32
            import torch.nn.functional as TF2
            dy = _bf(torch.randn(y.shape, generator=torch.Generator().manual_seed(1)))
            y.backward(dy.cuda())
            xr = xq.float().requires_grad_()
            wd, wu = _bf(st["lora_down.weight"]).float(), _bf(st["lora_up.weight"]).float()
            wb = _bf(w).float()
            if name.startswith("linear"):
                yr = TF2.linear(xr, wb) + TF2.linear(TF2.linear(xr, wd), wu) * it["scale"]
                yr.reshape(-1, yr.shape[-1]).backward(dy[:, : yr.shape[-1]].float())
                dxr = xr.grad.reshape(-1, xr.shape[-1])
            elif name == "conv3d":
                yr = TF2.conv3d(xr, wb, padding=(1, 0, 0)) + TF2.conv3d(TF2.conv3d(xr, wd, padding=(1, 0, 0)), wu) * it["scale"]
                yr.permute(0, 2, 3, 4, 1).reshape(dy.shape[0], -1).backward(dy[:, : yr.shape[1]].float())
                dxr = xr.grad.permute(0, 2, 3, 4, 1).reshape(xm.shape[0], -1)
            else:
                s_, p_ = (2 if name == "conv2d_s2" else 1), (1 if w.shape[2] == 3 else 0)
                yr = TF2.conv2d(xr, wb, stride=s_, padding=p_) + TF2.conv2d(TF2.conv2d(xr, wd, stride=s_, padding=p_), wu) * it["scale"]
                _cl(yr).backward(dy[:, : yr.shape[1]].float())
                dxr = _cl(xr.grad)
            assert relerr(xm.grad[:, : dxr.shape[1]].float(), dxr) < 3e-2, (name, "dx")
            # factor gradients of the fused node (dt rides in the backward-data launch; K-window of the rank columns for
            # windowed layers) against autograd of the reference expression
            wdr, wur = wd.clone().requires_grad_(), wu.clone().requires_grad_()
            xr2 = xq.float()
            if name.startswith("linear"):
                yr2 = TF2.linear(TF2.linear(xr2, wdr), wur) * it["scale"]
                yr2.reshape(-1, yr2.shape[-1]).backward(dy[:, : yr2.shape[-1]].float())
            elif name == "conv3d":
                yr2 = TF2.conv3d(TF2.conv3d(xr2, wdr, padding=(1, 0, 0)), wur) * it["scale"]
                yr2.permute(0, 2, 3, 4, 1).reshape(dy.shape[0], -1).backward(dy[:, : yr2.shape[1]].float())
            else:
                yr2 = TF2.conv2d(TF2.conv2d(xr2, wdr, stride=s_, padding=p_), wur) * it["scale"]
                _cl(yr2).backward(dy[:, : yr2.shape[1]].float())
            torch.cuda.synchronize()
            assert float(wdr.grad.norm()) > 0 and float(wur.grad.norm()) > 0
            assert relerr(mod.lora_up.weight.grad.float(), wur.grad) < 3e-2, (name, "dU")
            assert relerr(mod.lora_down.weight.grad.float(), wdr.grad) < 3e-2, (name, "dD")


@pytest.mark.parametrize("Np,Cp,taps,rp,grouped", [(320, 320, 1, 16, False), (72, 40, 9, 8, False), (128, 64, 3, 32, False),
                                                   (64, 128, 1, 16, True)])
def test_lora_merge_kernel(Np, Cp, taps, rp, grouped):
    """t2v_lora_merge: W_eff = W + s U D in the forward layout [n, tap*Cp + c] and the flipped-tap backward-data layout
    [c, (taps-1-tap)*Np + n], incl. ragged 64x64 tiles and a projection-group member (row/column block of wider buffers)."""
    import ctypes as C
    import t2v_amd.native as nv
    g = torch.Generator().manual_seed(Np + Cp + taps)
    K = taps * Cp
    W = torch.randn(Np, K, generator=g) * 0.05
    ldu = 3 * Np if grouped else Np
    Ufull = torch.randn(rp, ldu, generator=g) * 0.3
    c0 = Np if grouped else 0
    D = torch.randn(rp, K, generator=g) * 0.3
    s = 0.7
    Wd, Ud, Dd = W.cuda(), Ufull.cuda(), D.cuda()
    ldwf = K
    wf_full = torch.zeros(3 * Np if grouped else Np, K, dtype=torch.bfloat16, device="cuda")
    wb_full = torch.zeros(Cp, taps * (3 * Np if grouped else Np), dtype=torch.bfloat16, device="cuda")
    wf = wf_full[c0: c0 + Np]
    wb = wb_full[:, c0: c0 + Np] if grouped else wb_full
    jobs = (nv.LoraMergeJob * 1)()
    j = jobs[0]
    j.w32, j.up, j.ldu, j.down = Wd.data_ptr(), Ud.data_ptr() + 4 * c0, ldu, Dd.data_ptr()
    j.wf, j.ldwf, j.wb, j.ldwb = wf.data_ptr(), ldwf, wb.data_ptr(), wb_full.stride(0)
    j.Np, j.Cp, j.taps, j.rp, j.scale = Np, Cp, taps, rp, s
    total = nv.lib().t2v_lora_merge_plan(jobs, 1, None, 0)
    assert total == ((Np + 63) // 64) * ((Cp + 63) // 64) * taps
    tj = (C.c_int * total)()
    assert nv.lib().t2v_lora_merge_plan(jobs, 1, tj, total) == total
    jd = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).cuda()
    td = torch.frombuffer(bytearray(bytes(tj)), dtype=torch.int32).cuda()
    nv.call("t2v_lora_merge", jd.data_ptr(), 1, td.data_ptr(), total, nv.stream())
    torch.cuda.synchronize()
    ref = (W.double() + s * (Ufull[:, c0: c0 + Np].double().t() @ D.double())).float()
    refq = ref.to(torch.bfloat16)
    got = wf.cpu()
    d = (got.float() - refq.float()).abs()       # one rounding of the fp32 sum: only ties at the rounding boundary may differ
    assert float((d > 0).float().mean()) < 2e-3 and float(d.max()) <= float(ref.abs().max()) * 2.0 ** -7
    ref_b = got.view(Np, taps, Cp).flip(1).permute(2, 1, 0).reshape(Cp, taps * Np)
    assert torch.equal(wb.cpu(), ref_b)          # the backward layout carries exactly the same values
    if grouped:                                                            # nothing outside the member's blocks was touched
        assert float(wf_full[:c0].abs().sum()) == 0 and float(wf_full[c0 + Np:].abs().sum()) == 0
        assert float(wb_full[:, :c0].abs().sum()) == 0 and float(wb_full[:, c0 + Np:].abs().sum()) == 0


@pytest.mark.parametrize("kind", ["linear", "conv3x3", "conv3d", "conv_s2"])
def test_fused_lora_dropout_matches_the_unfused_masked_path(kind):
    """LoRA dropout inside the fused layer (the reference's default train mode, dropout_p = 0.1: utils/lora.py:35,49,89,119):
    the fused node (base + rank columns in one launch, masked rank update, masked dy for dt / dU) must reproduce the unfused
    composition `base(x) + dropout(up(down(x))) * scale` evaluated with the SAME counter-based mask — outputs, dx and both
    factor gradients."""
    from t2v_amd.functional import ConvCfg, LINEAR
    from t2v_amd.models import leaves
    from t2v_amd.models.leaves import run_layer
    from t2v_amd.training import FlatAdamW
    from t2v_amd.utils import lora as L
    import copy
    torch.manual_seed(3)
    if kind == "linear":
        mod, cfg, rows, cin = L.LoraInjectedLinear(320, 640, bias=True, r=16, dropout_p=0.1), LINEAR, 700, 320
    elif kind == "conv3x3":
        mod, cfg, rows, cin = L.LoraInjectedConv2d(64, 128, 3, 1, 1, r=16, dropout_p=0.1), ConvCfg.conv2d(3, 10, 12, 3, 1, 1), 360, 64
    elif kind == "conv3d":
        mod, cfg, rows, cin = L.LoraInjectedConv3d(64, 64, (3, 1, 1), (1, 0, 0), r=8, dropout_p=0.1), ConvCfg.conv3d_t(2, 6, 20), 240, 64
    else:
        mod, cfg, rows, cin = L.LoraInjectedConv2d(64, 64, 3, 2, 1, r=16, dropout_p=0.1), ConvCfg.conv2d(2, 8, 8, 3, 2, 1), 128, 64
    torch.nn.init.normal_(mod.lora_up.weight, std=0.1)
    ref = copy.deepcopy(mod).cuda().train()
    mod = mod.cuda().train()
    for m in (mod, ref):
        (m.linear if kind == "linear" else m.conv).requires_grad_(False)
    holder = torch.nn.Module(); holder.layer = mod
    opt = FlatAdamW([mod.lora_down.weight, mod.lora_up.weight], model=holder)
    holder_ref = torch.nn.Module(); holder_ref.layer = ref
    leaves.assign_dropout_names(holder_ref)           # same site name ("layer") -> same name-keyed dropout seed as the fused module
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(rows, cin, generator=g)).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    leaves.set_dropout_seed(77)
    ya = run_layer(mod, xa, cfg)                      # fused: bank entry attached, dropout active
    leaves.set_dropout_seed(77)
    yb = run_layer(ref, xb, cfg)                      # unfused three-launch composition, same seed
    assert ya.shape == yb.shape
    dy = _bf(torch.randn(ya.shape, generator=g)).cuda()
    opt.zero_grad()
    ya.backward(dy); yb.backward(dy)
    torch.cuda.synchronize()
    assert float(ya.float().abs().max()) > 0 and torch.isfinite(ya.float()).all()
    assert relerr(ya, yb) < 1e-2
    # the LoRA branch really is masked: ~10 % of the branch outputs are dropped
    assert relerr(xa.grad, xb.grad) < 2e-2
    assert relerr(mod.lora_up.weight.grad, ref.lora_up.weight.grad) < 2e-2
    assert relerr(mod.lora_down.weight.grad, ref.lora_down.weight.grad) < 2e-2
    # dropout really is on: the same layer with dropout off gives a different output
    mod.dropout.p = 0.0
    y0 = run_layer(mod, x.clone().requires_grad_(True), cfg)
    assert relerr(y0, yb) > 1e-3


def test_attention_deferred_rescale_branch():
    """The workgroup forward keeps its running maximum while no row's maximum grows by more than 2^6 and rescales O / l only
    then.  Random data rarely takes the rescale after the first block; this input FORCES it late and unevenly: a few key rows
    in the middle and at the end of the sequence are spiked against a few query rows (scores far above everything before),
    other rows never rescale.  Full-tensor comparison against the fp32 reference."""
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(17)
    nb, heads, S = 2, 2, 512
    C = heads * 64
    q = torch.randn(nb, S, C, generator=g); k = torch.randn(nb, S, C, generator=g); v = torch.randn(nb, S, C, generator=g)
    for (qi, ki, amp) in ((5, 300, 6.0), (5, 500, 9.0), (130, 70, 8.0), (257, 511, 7.0), (400, 33, 5.0)):
        k[:, ki, :64] = q[:, qi, :64] * amp / 8.0          # q.k / 8 ~ amp * |q|^2 / 64: dominates the row from key ki on
    q, k, v = _bf(q), _bf(k), _bf(v)
    ref = _sdpa_ref(q.float(), k.float(), v.float(), heads)
    o = F.attention(q.view(-1, C).cuda(), k.view(-1, C).cuda(), v.view(-1, C).cuda(), heads, F.SeqLayout(nb, S, S, 0, 1),
                    F.SeqLayout(nb, S, S, 0, 1))
    assert relerr(o, ref.view(-1, C)) < TOL
    assert float((o.float().cpu() - ref.view(-1, C)).abs().max()) < 0.06


# ----------------------------------------------------------------------------------------------------------------------
# 8-wave GEMM configurations (csrc/gemm_w8.hip) through the C ABI: every configuration, column steps, in-launch split-K
def _keep_plane(keep):
    """The keep-bit plane of T2VGemm.lr_plane for a bool mask [M, W] (W % 32 == 0), restated from include/t2v_abi.h: 16-bit word
    (j, row, h) at halfword (j * M + row) * 2 + h; bit b = keep of column 32 j + 8 (b >> 2) + 4 h + (b & 3)."""
    import numpy as np
    M, W = keep.shape
    k = keep.numpy().astype(np.uint32).reshape(M, W // 32, 4, 2, 4)          # [row, j, b >> 2, h, b & 3]
    words = np.zeros((W // 32, M, 2), dtype=np.uint32)
    for q in range(4):
        for e in range(4):
            words |= k[:, :, q, :, e].transpose(1, 0, 2) << (4 * q + e)
    return torch.from_numpy(words.astype(np.uint16).view(np.int16).reshape(-1).copy())


def _w8_problem(M, N, rc, K, taps, res, seed=0):
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    g = torch.Generator().manual_seed(seed)
    cin = K // taps
    geom, x4 = None, None
    if taps == 9:
        nimg, side = 2, int((M // 2) ** 0.5)
        assert nimg * side * side == M
        geom = F.ConvCfg.conv2d(nimg, side, side, 3, 1, 1).fwd_geom(cin)
    elif taps == 3:
        geom = F.ConvCfg.conv3d_t(1, 4, M // 4).fwd_geom(cin)
    a = _bf(torch.randn(M, cin, generator=g)).cuda()
    w = _bf(torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = _bf(torch.randn(M, N, generator=g)).cuda() if res else None
    d = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    kw = dict(M=M, N=N + rc, K=K, A=a.data_ptr(), lda=cin, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, bias=b.data_ptr(),
              a_mode=1 if geom is not None else 0, geom=geom, R=nv.ptr(r), ldr=N if res else 0)
    keep = [a, w, b, r, d]
    t = None
    if rc:
        w2 = _bf(torch.randn(rc, K, generator=g) * K ** -0.5).cuda()
        t = torch.empty(M, rc, dtype=torch.bfloat16, device="cuda")
        kw.update(B2=w2.data_ptr(), ldb2=K, n_split=N, D2=t.data_ptr(), ldd2=rc)
        keep += [w2, t]
    return kw, d, t, keep


@pytest.mark.parametrize("M,N,rc,K,taps,res", [(512, 320, 16, 320, 1, 1), (1152, 640, 16, 1920, 3, 0), (1152, 320, 16, 2880, 9, 1),
                                                (300, 1280, 48, 1280, 1, 0), (2048, 128, 0, 1152, 9, 0)])
def test_w8_configurations_match_the_table_kernels(M, N, rc, K, taps, res):
    """Every 8-wave configuration x (whole-BN / partial column step) x (no split, 2, 3 K splits) against the 4-wave kernel the
    heuristic picks for the same descriptor (itself checked against torch above).  K groups and splits change the summation
    order: tolerance 1e-2 of the output scale; configurations without either must be bit-identical."""
    import ctypes as C
    import os
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    kw, d, t, keep = _w8_problem(M, N, rc, K, taps, res, seed=M + K)
    desc = F.make_gemm(**kw)
    os.environ["T2V_GEMM_FORCE_CFG"] = "0,2,1"            # 128x128 4-wave tile: the reference for this test
    try:
        nv.call("t2v_gemm", C.byref(desc), nv.stream())
    finally:
        del os.environ["T2V_GEMM_FORCE_CFG"]
    torch.cuda.synchronize()
    ref, reft = d.float().clone(), (t.float().clone() if rc else None)
    scale = float(ref.abs().max())
    lib = nv.lib()
    ncfg = lib.t2v_gemm_w8_configs()
    assert ncfg >= 17
    for cfg in range(ncfg):
        if cfg == 8:
            continue                                      # (register-bound experiment, not in the production set)
        for nstep, splits in ((0, 1), (160, 1), (0, 2), (96, 3)):
            d.zero_()
            if rc:
                t.zero_()
            nv.call("t2v_gemm_w8", C.byref(desc), cfg, nstep, splits, nv.stream())
            torch.cuda.synchronize()
            err = float((d.float() - ref).abs().max()) / scale
            if rc:
                err = max(err, float((t.float() - reft).abs().max()) / float(reft.abs().max()))
            assert err < 1e-2, (cfg, nstep, splits, err)
    ws = F._gemm_workspace()
    assert int(ws[:16384].view(torch.int32).abs().max()) == 0, "split-K counters must be left zero"


@pytest.mark.parametrize("M,N,K,bias,rb,act,res", [(300, 320, 320, 0, 0, 0, 0), (300, 320, 320, 0, 1, 1, 1), (1000, 640, 640, 1, 1, 1, 0),
                                                    (256, 200, 320, 0, 0, 1, 1), (2048, 1280, 1280, 1, 1, 0, 1)])
def test_w8_register_epilogue_operand_combinations(M, N, K, bias, rb, act, res):
    """The register epilogue of the 8-wave kernels requests bias / row-bias / residual with UNCONDITIONAL loads two chunks ahead (an
    absent operand reads a zero page, lanes outside the tile read row / column 0 — gemm_w8.hip, round 6): every combination of
    present / absent operands, with SiLU, on ragged row counts and a column count that is not a whole tile, against fp32 torch on the
    bf16 operands (the epilogue of ops/conv.py:forward: act(x W^T + bias + rowbias[image]) + residual)."""
    import ctypes as C
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    g = torch.Generator().manual_seed(M + N + 7 * bias + 3 * rb)
    a = _bf(torch.randn(M, K, generator=g)).cuda()
    w = _bf(torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    nimg = 4 if M % 4 == 0 else 3
    rows_per = M // nimg if M % nimg == 0 else M
    rbt = _bf(torch.randn(M // rows_per, N, generator=g)).cuda() if rb else None
    r = _bf(torch.randn(M, N, generator=g)).cuda() if res else None
    d = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    kw = dict(M=M, N=N, K=K, A=a.data_ptr(), lda=K, B=w.data_ptr(), ldb=K, D=d.data_ptr(), ldd=N, bias=nv.ptr(b), a_mode=0, geom=None,
              rowbias=nv.ptr(rbt), ldrb=N if rb else 0, rows_per_rb=rows_per if rb else 0, R=nv.ptr(r), ldr=N if res else 0,
              act=1 if act else 0)
    desc = F.make_gemm(**kw)
    ref = a.float() @ w.float().T
    if bias:
        ref = ref + b
    if rb:
        ref = ref + rbt.float().repeat_interleave(rows_per, 0)
    if act:
        ref = torch.nn.functional.silu(ref)
    if res:
        ref = ref + r.float()
    scale = float(ref.abs().max())
    for cfg in (12, 13, 14, 16, 17, 18, 19, 20, 21, 22):          # the configurations with a register epilogue
        for nstep, splits in ((0, 1), (160, 1), (0, 2)):
            d.fill_(float("nan"))
            nv.call("t2v_gemm_w8", C.byref(desc), cfg, nstep, splits, nv.stream())
            torch.cuda.synchronize()
            assert torch.isfinite(d.float()).all(), (cfg, nstep, splits)
            err = float((d.float() - ref).abs().max()) / scale
            assert err < 1.2e-2, (cfg, nstep, splits, err)
    ws = F._gemm_workspace()
    assert int(ws[:16384].view(torch.int32).abs().max()) == 0, "split-K counters must be left zero"


@pytest.mark.parametrize("M,N,K,taps,rp,res,p_drop", [(512, 320, 320, 1, 16, 1, 0.1), (1152, 640, 1920, 3, 16, 0, 0.1),
                                                       (1152, 320, 2880, 9, 16, 1, 0.1), (300, 1280, 1280, 1, 32, 0, 0.25),
                                                       (2048, 128, 1152, 9, 8, 0, 0.1), (256, 2560, 320, 1, 24, 0, 0.0)])
def test_w8_rank_epilogue_term_forward(M, N, K, taps, rp, res, p_drop):
    """T2VGemm.lr_mode 2 — the dropped LoRA branch inside the base layer's launch (utils/lora.py:57-62 with the dropout of :49):
    y = x (*) W^T + bias + R + s mask (t U^T), t = bf16(x (*) D^T) computed by every column tile and saved to D2.  Against torch
    fp32 on the bf16 operands with the protocol mask, over the LR configurations x column steps x K splits, with and without
    column statistics (the staged epilogue)."""
    import ctypes as C
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    from oracle.dropout import keep_mask
    kw, d, _, keep = _w8_problem(M, N, 0, K, taps, res, seed=M + K + rp)
    a, w, b, r = keep[0], keep[1], keep[2], keep[3]
    g = torch.Generator().manual_seed(77 + rp)
    rk = 16 if rp <= 16 else 32
    Dw = _bf(torch.randn(rp, K, generator=g) * K ** -0.5).cuda()
    UT = torch.zeros(N, rk); UT[:, :rp] = torch.randn(N, rp, generator=g) * 0.5
    UT = _bf(UT).cuda()
    t = torch.zeros(M, rp, dtype=torch.bfloat16, device="cuda")
    s, seed = 0.7, 0x5EED1234
    kw.update(B2=Dw.data_ptr(), ldb2=K, D2=t.data_ptr(), ldd2=rp,
              lr=dict(mode=2, rp=rp, b=UT.data_ptr(), ldb=rk, scale=s, drop_p=p_drop, drop_seed=seed))
    # reference: the same launch without the term, then the branch in fp32
    kw0 = {k: v for k, v in kw.items() if k not in ("lr", "B2", "ldb2", "D2", "ldd2")}
    kw0.update(B2=Dw.data_ptr(), ldb2=K, n_split=N, D2=t.data_ptr(), ldd2=rp, N=N + rp)
    d.zero_()
    nv.call("t2v_gemm", C.byref(F.make_gemm(**{**kw0, "R": None, "ldr": 0})), nv.stream())
    torch.cuda.synchronize()
    base32_src = None
    # fp32 base from torch directly (the bf16-rounded launch output would double-round)
    af = a.float().cpu()
    if taps == 1:
        cols = af
    elif taps == 9:
        side = int((M // 2) ** 0.5)
        x4 = af.view(2, side, side, -1)
        xp = torch.zeros(2, side + 2, side + 2, x4.shape[-1]); xp[:, 1:-1, 1:-1] = x4
        cols = torch.cat([xp[:, ky:ky + side, kx:kx + side] for ky in range(3) for kx in range(3)], -1).reshape(M, -1)
    else:
        x4 = af.view(1, 4, M // 4, -1)
        xp = torch.zeros(1, 6, M // 4, x4.shape[-1]); xp[:, 1:-1] = x4
        cols = torch.cat([xp[:, ky:ky + 4] for ky in range(3)], -1).reshape(M, -1)
    base = cols @ w.float().cpu().T + b.cpu()
    tref = _bf(cols @ Dw.float().cpu().T)
    m = keep_mask(seed, M, N, p_drop).float() / (1.0 - p_drop) if p_drop > 0 else torch.ones(M, N)
    ref = base + (r.float().cpu() if res else 0) + s * m * (tref.float() @ UT.float().cpu()[:, :rp].T)
    assert relerr(t, tref) < 1e-2          # (the unfused launch's t: same rounding point)
    scale = float(ref.abs().max())
    # (round 6) the launch also leaves the keep bits of its mask for the backward-data launch (T2VGemm.lr_plane)
    plane = pref = None
    if p_drop > 0 and N % 32 == 0:
        plane = torch.zeros(M * N // 16, dtype=torch.int16, device="cuda")
        pref = _keep_plane(keep_mask(seed, M, N, p_drop))
        kw["lr"]["plane"] = plane.data_ptr()
    desc = F.make_gemm(**kw)
    assert nv.lib().t2v_gemm_lr_ok(C.byref(desc)) == 1
    for cfg in (12, 17, 21, 22, 14, 16, 19, 20):      # (14, 16, 19, 20: K-group configurations, rank phase after the groups met)
        for nstep, splits in ((0, 1), (160, 1), (0, 2), (96, 3)):
            d.zero_(); t.zero_()
            if plane is not None:
                plane.zero_()
            nv.call("t2v_gemm_w8", C.byref(desc), cfg, nstep, splits, nv.stream())
            torch.cuda.synchronize()
            err = float((d.float().cpu() - ref).abs().max()) / scale
            et = relerr(t, tref)
            assert err < 1.2e-2 and et < 1e-2, (cfg, nstep, splits, err, et)
            if plane is not None:
                assert torch.equal(plane.cpu(), pref), (cfg, nstep, splits, "keep-bit plane")
    # through t2v_gemm (heuristic configuration), with the GroupNorm column statistics of the FINAL output
    if N % 32 == 0 and M % 128 == 0:
        d.zero_()
        import os
        os.environ["T2V_GEMM_FORCE_CFG"] = "117,10,1"     # a configuration whose staged epilogue can emit them with the term
        try:
            info = F.launch_gemm(cs={"mode": 1}, **kw)
        finally:
            del os.environ["T2V_GEMM_FORCE_CFG"]
        assert info is not None
        buf, bm, mm, nb = info
        G, nd = 32, 1
        sums = torch.empty(nd * G * 2, device="cuda"); refs = torch.empty_like(sums)
        nv.call("t2v_gn_finish", buf.data_ptr(), nd, M, N, G, sums.data_ptr(), nv.stream())
        ws = F._gn_workspace(nd, G, d.device)
        nv.call("t2v_gn_stats", d.data_ptr(), N, nd, M, N, G, refs.data_ptr(), ws.data_ptr(), nv.stream())
        torch.cuda.synchronize()
        assert float((d.float().cpu() - ref).abs().max()) / scale < 1.2e-2
        assert relerr(sums, refs) < 1e-5
        d.zero_()                                          # and through whatever t2v_gemm selects on its own
        F.launch_gemm(cs={"mode": 1}, **kw)
        torch.cuda.synchronize()
        assert float((d.float().cpu() - ref).abs().max()) / scale < 1.2e-2
    ws = F._gemm_workspace()
    assert int(ws[:16384].view(torch.int32).abs().max()) == 0, "split-K counters must be left zero"


@pytest.mark.parametrize("M,N,K,taps,rp,res,masked", [(512, 320, 320, 1, 16, 0, 0), (1152, 640, 1920, 3, 16, 0, 0),
                                                       (1152, 320, 2880, 9, 16, 1, 0), (300, 1280, 1280, 1, 32, 0, 1),
                                                       (2048, 128, 1152, 9, 8, 0, 0), (512, 640, 640, 1, 24, 1, 1)])
def test_w8_rank_epilogue_term_from_memory(M, N, K, taps, rp, res, masked):
    """T2VGemm.lr_mode 1 — D += sum_tap LA[src(m, tap)] LB[n, tap]^T with LA read from memory: the backward-data launch of a
    wrapped layer (dx = dy (*) W^T + s dt (*) D^T: windowed, no mask) and the masked one-tap form."""
    import ctypes as C
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    from oracle.dropout import keep_mask
    kw, d, _, keep = _w8_problem(M, N, 0, K, taps, res, seed=M + K + rp + 1)
    a, w, b, r = keep[0], keep[1], keep[2], keep[3]
    g = torch.Generator().manual_seed(99 + rp)
    rk = 16 if rp <= 16 else 32
    LA = _bf(torch.randn(M, rp, generator=g)).cuda()
    LB = torch.zeros(N, taps, rk); LB[:, :, :rp] = torch.randn(N, taps, rp, generator=g) * 0.3
    LB = _bf(LB.reshape(N, taps * rk)).cuda()
    p_drop, seed, s = (0.1, 0xC0FFEE, 0.6) if masked else (0.0, 0, 1.0)
    kw.update(lr=dict(mode=1, rp=rp, taps=taps, a=LA.data_ptr(), lda=rp, b=LB.data_ptr(), ldb=taps * rk, scale=s, drop_p=p_drop,
                      drop_seed=seed))
    af, laf = a.float().cpu(), LA.float().cpu()

    def gather(src):
        if taps == 1:
            return src
        if taps == 9:
            side = int((M // 2) ** 0.5)
            x4 = src.view(2, side, side, -1)
            xp = torch.zeros(2, side + 2, side + 2, x4.shape[-1]); xp[:, 1:-1, 1:-1] = x4
            return torch.cat([xp[:, ky:ky + side, kx:kx + side] for ky in range(3) for kx in range(3)], -1).reshape(M, -1)
        x4 = src.view(1, 4, M // 4, -1)
        xp = torch.zeros(1, 6, M // 4, x4.shape[-1]); xp[:, 1:-1] = x4
        return torch.cat([xp[:, ky:ky + 4] for ky in range(3)], -1).reshape(M, -1)

    base = gather(af) @ w.float().cpu().T + b.cpu()
    lbf = LB.float().cpu().view(N, taps, rk)[:, :, :rp].reshape(N, taps * rp)
    term = gather(laf) @ lbf.T
    m = keep_mask(seed, M, N, p_drop).float() / (1.0 - p_drop) if masked else 1.0
    ref = base + (r.float().cpu() if res else 0) + s * m * term
    scale = float(ref.abs().max())
    desc = F.make_gemm(**kw)
    assert nv.lib().t2v_gemm_lr_ok(C.byref(desc)) == 1
    for cfg in (12, 17, 21, 22, 14, 16, 19, 20):
        for nstep, splits in ((0, 1), (160, 1), (0, 2), (96, 3)):
            d.zero_()
            nv.call("t2v_gemm_w8", C.byref(desc), cfg, nstep, splits, nv.stream())
            torch.cuda.synchronize()
            err = float((d.float().cpu() - ref).abs().max()) / scale
            assert err < 1.2e-2, (cfg, nstep, splits, err)
    d.zero_()
    nv.call("t2v_gemm", C.byref(desc), nv.stream())
    torch.cuda.synchronize()
    assert float((d.float().cpu() - ref).abs().max()) / scale < 1.2e-2


@pytest.mark.parametrize("M,N,K,members,rpe,res,p_drop", [(512, 320, 320, 1, 16, 0, 0.1), (1152, 640, 1920, 3, 16, 0, 0.1),
                                                            (300, 1280, 1280, 1, 32, 1, 0.25), (640, 320, 2560, 1, 8, 0, 0.1),
                                                            (512, 640, 1280, 2, 32, 0, 0.1), (256, 1280, 320, 1, 24, 1, 0.0),
                                                            (384, 320, 640, 2, 16, 0, 0.1)])
def test_w8_rank_epilogue_term_dt_in_launch(M, N, K, members, rpe, res, p_drop):
    """T2VGemm.lr_mode 3 (round 6) — the backward-data launch of a dropped linear wrapper / projection group forms dt itself:
    dt = bf16(1/(1-p) (mask dy) U^T) from rank fragments whose MFMAs take masked A fragments (member i of a group = K block i with
    its own seed and mask width), written to D2, and D = dy W^T + bias + R + dt LB^T.  Against torch fp32 on the bf16 operands with
    the protocol masks (oracle/dropout.py), over the LR configurations x column steps x K splits, and through t2v_gemm with the
    column statistics of the stored output (staged epilogue)."""
    import ctypes as C
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    from oracle.dropout import keep_mask
    kw, d, _, keep = _w8_problem(M, N, 0, K, 1, res, seed=M + K + rpe + members)
    a, w, b, r = keep[0], keep[1], keep[2], keep[3]
    g = torch.Generator().manual_seed(123 + rpe + members)
    rp, W = rpe * members, K // members
    rk = 16 * ((rp + 15) // 16)
    U = torch.zeros(rp, K)
    for i in range(members):                        # block diagonal, zeros stored (lora_bank's group layout)
        U[i * rpe:(i + 1) * rpe, i * W:(i + 1) * W] = torch.randn(rpe, W, generator=g) * W ** -0.5
    U = _bf(U).cuda()
    LB = torch.zeros(N, rk); LB[:, :rp] = torch.randn(N, rp, generator=g) * 0.3
    LB = _bf(LB).cuda()
    dt = torch.zeros(M, rp, dtype=torch.bfloat16, device="cuda")
    seeds = [0xABCD01, 0x1234567, 0x7654321][:members]
    kw.update(B2=U.data_ptr(), ldb2=K, D2=dt.data_ptr(), ldd2=rp,
              lr=dict(mode=3, rp=rp, b=LB.data_ptr(), ldb=rk, drop_p=p_drop, drop_seed=seeds[0],
                      group_cols=W if members > 1 else 0, group_seeds=seeds[1:]))
    af = a.float().cpu()
    if p_drop > 0:
        m = torch.cat([keep_mask(seeds[i], M, W, p_drop).float() for i in range(members)], 1) / (1.0 - p_drop)
    else:
        m = torch.ones(M, K)
    dtref = _bf((m * af) @ U.float().cpu().T)
    ref = af @ w.float().cpu().T + b.cpu() + (r.float().cpu() if res else 0) + dtref.float() @ LB.float().cpu()[:, :rp].T
    scale = float(ref.abs().max())
    desc = F.make_gemm(**kw)
    assert nv.lib().t2v_gemm_lr_ok(C.byref(desc)) == 1
    descs = [("hash", desc)]
    if p_drop > 0 and W % 64 == 0:
        # the same launch reading the keep bits from the plane the forward launch left (member i at byte i * M * W / 8); WRONG seeds
        # prove that the bits, not a hash, decide
        plane = torch.cat([_keep_plane(keep_mask(seeds[i], M, W, p_drop)) for i in range(members)]).cuda()
        kwp = dict(kw)
        kwp["lr"] = dict(kw["lr"], plane=plane.data_ptr(), drop_seed=1, group_seeds=[2, 3][:members - 1])
        descs.append(("plane", F.make_gemm(**kwp)))
    for how, dsc in descs:
        for cfg in (12, 17, 21, 22, 14, 16, 19, 20):
            for nstep, splits in ((0, 1), (160, 1), (0, 2), (96, 3)):
                d.zero_(); dt.zero_()
                nv.call("t2v_gemm_w8", C.byref(dsc), cfg, nstep, splits, nv.stream())
                torch.cuda.synchronize()
                err = float((d.float().cpu() - ref).abs().max()) / scale
                et = relerr(dt, dtref)
                assert err < 1.2e-2 and et < 1e-2, (how, cfg, nstep, splits, err, et)
    d.zero_(); dt.zero_()
    nv.call("t2v_gemm", C.byref(desc), nv.stream())        # whatever t2v_gemm selects on its own
    torch.cuda.synchronize()
    assert float((d.float().cpu() - ref).abs().max()) / scale < 1.2e-2 and relerr(dt, dtref) < 1e-2
    if N % 32 == 0 and M % 128 == 0:                        # staged epilogue: column statistics of the FINAL output ride along
        d.zero_(); dt.zero_()
        import os
        os.environ["T2V_GEMM_FORCE_CFG"] = "117,10,1"
        try:
            info = F.launch_gemm(cs={"mode": 1}, **kw)
        finally:
            del os.environ["T2V_GEMM_FORCE_CFG"]
        assert info is not None
        buf, bm, mm, nb = info
        G, nd = 32, 1
        sums = torch.empty(nd * G * 2, device="cuda"); refs = torch.empty_like(sums)
        nv.call("t2v_gn_finish", buf.data_ptr(), nd, M, N, G, sums.data_ptr(), nv.stream())
        ws = F._gn_workspace(nd, G, d.device)
        nv.call("t2v_gn_stats", d.data_ptr(), N, nd, M, N, G, refs.data_ptr(), ws.data_ptr(), nv.stream())
        torch.cuda.synchronize()
        assert float((d.float().cpu() - ref).abs().max()) / scale < 1.2e-2 and relerr(dt, dtref) < 1e-2
        assert relerr(sums, refs) < 1e-5
    ws = F._gemm_workspace()
    assert int(ws[:16384].view(torch.int32).abs().max()) == 0, "split-K counters must be left zero"


@pytest.mark.parametrize("force", ["114,0,1", "113,0,1", "2,2,1", "0,0,1", "112,5,2"])
@pytest.mark.parametrize("M,C_,K,taps,rpd,res", [(1024, 320, 960, 3, 256, 1), (2048, 640, 640, 1, 1024, 0), (512, 320, 2880, 9, 256, 1)])
def test_gemm_epilogue_groupnorm_statistics(M, C_, K, taps, rpd, res, force):
    """T2VGemm.colsum: the GroupNorm sums a GEMM epilogue leaves behind + t2v_gn_finish against the standalone statistics
    kernels on the stored output — forward (sum y, sum y^2) and backward (sum dz*gamma, sum dz*gamma*xh, SiLU on and off)."""
    import ctypes as C
    import t2v_amd.functional as F
    import t2v_amd.native as nv
    G = 32
    nd = M // rpd
    lib = nv.lib()
    # pin the kernel: 8-wave configurations (incl. a column step + K split) and 4-wave tiles of gemm.hip
    import os
    os.environ["T2V_GEMM_FORCE_CFG"] = force
    try:
        kw, d, t, keep = _w8_problem(M, C_, 16, K, taps, res, seed=7 + M)
        # ---- mode 1
        info = F.launch_gemm(cs={"mode": 1}, **kw)
        assert info is not None, "the pinned 8-wave configuration must report column statistics"
        buf, bm, mm, nb = info
        assert mm == M and nb == C_ and rpd % bm == 0
        sums = torch.empty(nd * G * 2, device="cuda")
        nv.call("t2v_gn_finish", buf.data_ptr(), nd, rpd, C_, G, sums.data_ptr(), nv.stream())
        ref = torch.empty_like(sums)
        ws = F._gn_workspace(nd, G, d.device)
        nv.call("t2v_gn_stats", d.data_ptr(), C_, nd, rpd, C_, G, ref.data_ptr(), ws.data_ptr(), nv.stream())
        torch.cuda.synchronize()
        assert relerr(sums, ref) < 1e-5
        # ---- mode 2: the GEMM output plays dL/d(norm output); x, gamma, beta of the norm are independent tensors
        g = torch.Generator().manual_seed(11)
        x = _bf(torch.randn(M, C_, generator=g) * 2 + 0.5).cuda()
        gamma = (torch.randn(C_, generator=g) * 0.5 + 1).cuda()
        beta = (torch.randn(C_, generator=g) * 0.3).cuda()
        fs = torch.empty(nd * G * 2, device="cuda")
        nv.call("t2v_gn_stats", x.data_ptr(), C_, nd, rpd, C_, G, fs.data_ptr(), ws.data_ptr(), nv.stream())
        kw2 = dict(kw)
        kw2["R"], kw2["ldr"] = None, 0
        for silu, dp, dseed in ((1, 0.0, 0), (0, 0.0, 0), (1, 0.1, 0xD50F)):      # (last: a norm that drops behind its SiLU)
            info = F.launch_gemm(cs={"mode": 2, "x": x.data_ptr(), "ldx": C_, "sums": fs.data_ptr(), "gamma": gamma.data_ptr(),
                                     "beta": beta.data_ptr(), "eps": 1e-5, "G": G, "silu": silu, "domain_rows": rpd,
                                     "drop_p": dp, "drop_seed": dseed}, **kw2)
            assert info is not None
            bs = torch.empty(nd * G * 2, device="cuda")
            nv.call("t2v_gn_finish", info[0].data_ptr(), nd, rpd, C_, G, bs.data_ptr(), nv.stream())
            bref = torch.empty_like(bs)
            nv.call("t2v_gn_bwd_stats", x.data_ptr(), C_, d.data_ptr(), C_, nd, rpd, C_, G, fs.data_ptr(), gamma.data_ptr(),
                    beta.data_ptr(), 1e-5, silu, dp, dseed, bref.data_ptr(), ws.data_ptr(), None, None, None, nv.stream())
            torch.cuda.synchronize()
            assert relerr(bs, bref) < 1e-4, (silu, dp)
    finally:
        del os.environ["T2V_GEMM_FORCE_CFG"]


def test_groupnorm_fusion_end_to_end_matches_the_unfused_ops():
    """resnet-like chain conv -> GroupNorm(SiLU) -> conv with the statistics riding in the GEMM epilogues (forward and
    backward) against the same chain with T2V_GN_FUSE off: same outputs and input gradients to bf16 rounding."""
    import os
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(3)
    nimg, H, W, Cc = 4, 16, 16, 320
    x0 = _bf(torch.randn(nimg * H * W, Cc, generator=g))
    w1 = torch.randn(Cc, Cc, 3, 3, generator=g) * (9 * Cc) ** -0.5
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) * (9 * Cc) ** -0.5
    gam, bet = torch.randn(Cc, generator=g) * 0.3 + 1, torch.randn(Cc, generator=g) * 0.2
    dy = _bf(torch.randn(nimg * H * W, Cc, generator=g))
    cfg = F.ConvCfg.conv2d(nimg, H, W, 3, 1, 1)
    os.environ["T2V_GEMM_FORCE_CFG"] = "114,0,1"
    outs = []
    try:
        for fuse in (True, False):
            F._cs_enabled = fuse
            xd = _dev(x0)
            ws = [_dev(w1, False), _dev(w2, False)]
            gd, bd = _dev(gam, False), _dev(bet, False)
            h = F.conv_linear(xd, ws[0], None, cfg, colsum=True)
            assert hasattr(h, "_t2v_cs") == fuse
            a = F.group_norm(h, gd, bd, 32, 1e-5, True, nimg)
            y = F.conv_linear(a, ws[1], None, cfg)
            y.backward(dy.cuda())
            torch.cuda.synchronize()
            outs.append((y.detach().float().cpu(), xd.grad.float().cpu()))
    finally:
        F._cs_enabled = os.environ.get("T2V_GN_FUSE", "1") != "0"
        del os.environ["T2V_GEMM_FORCE_CFG"]
    assert relerr(outs[0][0], outs[1][0]) < 5e-3
    assert relerr(outs[0][1], outs[1][1]) < 5e-3


@pytest.mark.parametrize("groups,rpg,cols", [(2, 16384, 320), (2, 77, 1280), (32, 1024, 640), (1, 5, 8), (4, 4096, 1280)])
def test_rowgroup_sum(groups, rpg, cols):
    """t2v_rowgroup_sum: the gradient of a per-video row-bias (time embedding) / of keys and values shared by a video's frames —
    fp32 accumulation in a fixed order (two launches of the same input give the same bits)."""
    import t2v_amd.functional as F
    g = torch.Generator().manual_seed(groups + rpg + cols)
    x = _bf(torch.randn(groups * rpg, cols, generator=g))
    ref = x.float().view(groups, rpg, cols).sum(1)
    xd = x.cuda()
    y1, y2 = F.rowgroup_sum(xd, groups, rpg), F.rowgroup_sum(xd, groups, rpg)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert relerr(y1, ref) < 5e-3
