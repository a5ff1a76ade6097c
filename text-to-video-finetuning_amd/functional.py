"""Autograd-aware operators of the native path: thin torch.autograd.Function wrappers around the C ABI.

Activations are bf16 "token matrices" `[rows, C]` (channels-last; rows = images*H*W, unit inner stride,
row stride = `ld`).  Every op here launches hand-written HIP kernels through `native.call`; nothing
falls back to torch math on the device path (torch is used for allocation, views and the tiny
reductions of bias-like gradients).

Reference call sites replaced (all un-vendored diffusers leaves, see include/t2v_abi.h):
nn.Linear / nn.Conv2d / nn.Conv3d forward+backward, F.group_norm(+silu), F.layer_norm,
F.scaled_dot_product_attention (AttnProcessor2_0, train.py:138-139), GEGLU, SiLU.
"""
import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import native as nv

BF16 = torch.bfloat16


# --------------------------------------------------------------------------- helpers
def _mat(t, name="tensor"):
    if t.dim() != 2 or t.dtype != BF16 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"t2v_amd: {name} must be a 2-D bf16 matrix with unit inner stride, got "
                           f"{tuple(t.shape)} {t.dtype} strides {t.stride()}")
    nv.require_cuda(t)
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def ceil8(n):
    return (n + 7) // 8 * 8


@dataclass(frozen=True)
class ConvCfg:
    """Geometry of one sliding-window layer.  kind: 'linear' | 'conv' (2-D window over an (nimg,H,W) grid;
    the (3,1,1) temporal Conv3d is the window KH=3,KW=1 over the grid (B, F, H*W))."""
    kind: str = "linear"
    nimg: int = 1
    H: int = 1           # real source grid
    W: int = 1
    KH: int = 1
    KW: int = 1
    stride: int = 1
    pad_y: int = 0
    pad_x: int = 0
    up: int = 0          # 1: source is nearest-2x upsampled on the fly (Upsample2D)
    Ho: int = 1
    Wo: int = 1

    @staticmethod
    def conv2d(nimg, H, W, k=3, stride=1, pad=1, up=0, Ho=None, Wo=None):
        Hv, Wv = H << up, W << up
        if Ho is None:
            Ho = (Hv + 2 * pad - k) // stride + 1
            Wo = (Wv + 2 * pad - k) // stride + 1
        return ConvCfg("conv", nimg, H, W, k, k, stride, pad, pad, up, Ho, Wo)

    @staticmethod
    def conv3d_t(B, F, HW):
        return ConvCfg("conv", B, F, HW, 3, 1, 1, 1, 0, 0, F, HW)

    def taps(self):
        return self.KH * self.KW

    def fwd_geom(self, C_in):
        return nv.ConvGeom(C_in, self.H << self.up, self.W << self.up, self.Ho, self.Wo, self.KH, self.KW, self.stride,
                           self.stride, self.pad_y, self.pad_x, 1, self.up)

    def bwd_geom(self, C_out):
        # gather over dY (grid Ho x Wo) producing dX on the virtual input grid; taps are flipped in the prepared weight
        return nv.ConvGeom(C_out, self.Ho, self.Wo, self.H << self.up, self.W << self.up, self.KH, self.KW, 1, 1,
                           self.KH - 1 - self.pad_y, self.KW - 1 - self.pad_x, self.stride, 0)


LINEAR = ConvCfg()


# --------------------------------------------------------------------------- weight preparation (bf16 GEMM layouts)
# Parameters that train are updated by this library's own kernels (t2v_adamw on the flat buffer, inside replayed graphs): torch's
# version counters never move.  Every update path bumps this epoch instead (FlatAdamW.step / load_state_dict); caches of values
# DERIVED from trainable parameters outside the trainer's own refresh (folded LoRA weights of the forward-only temporal unit, the
# sampler's captured UNet call) carry it in their tags.
weights_epoch = [0]


def note_weights_changed():
    weights_epoch[0] += 1


def clear_weight_cache(model=None):
    """Drop cached prepared copies (they live on the Parameter objects themselves, so they die with the model) and the folded
    LoRA weights of the no-grad forward (models/leaves.py: `W + s up down` per wrapper and the fused temporal units' [3C, C] /
    permuted Wo copies, cached on the modules — ~2 bytes per UNet weight after a sampling pass with wrappers in place)."""
    if model is not None:
        for p in model.parameters():
            p.__dict__.pop("_t2v_prep", None)
        for m in model.modules():
            m.__dict__.pop("_t2v_folded", None)
            m.__dict__.pop("_t2v_fused_w", None)
            m.__dict__.pop("_t2v_nograd_w", None)


def _prep_compute(w, kind, cfg):
    """fwd: [Np, Kp] with K ordered (tap, c);  bwd: [Kin_p, taps_flipped*Np]."""
    wd = w.detach()
    if wd.dim() == 2:
        n, k = wd.shape
        w3 = wd.reshape(n, 1, k)                      # [N, taps=1, Cin]
    elif wd.dim() == 4:                               # Conv2d [Co, Ci, KH, KW]
        n, k = wd.shape[0], wd.shape[1]
        w3 = wd.permute(0, 2, 3, 1).reshape(n, -1, k)
    elif wd.dim() == 5:                               # Conv3d [Co, Ci, 3, 1, 1]
        n, k = wd.shape[0], wd.shape[1]
        w3 = wd[:, :, :, 0, 0].permute(0, 2, 1)
    else:
        raise RuntimeError(f"t2v_amd: unsupported weight rank {wd.dim()}")
    taps = w3.shape[1]
    npad, kpad = ceil8(n), ceil8(k)
    if npad != n or kpad != k:
        wp = torch.zeros(npad, taps, kpad, dtype=wd.dtype, device=wd.device)
        wp[:n, :, :k] = w3
        w3 = wp
    if kind == "fwd32":      # fp32 master in forward GEMM layout (input of the LoRA merge kernel)
        return w3.reshape(npad, taps * kpad).float().contiguous()
    if kind == "fwd":
        return w3.reshape(npad, taps * kpad).to(BF16).contiguous()
    # bwd-data: Wb[ci][tap'][co] = W[co][taps-1-tap'][ci]   (flip of the (KH,KW) window == reversal of the flat tap index)
    return w3.flip(1).permute(2, 1, 0).reshape(kpad, taps * npad).to(BF16).contiguous()


def resync_prepared(params):
    """Captured graphs keep reading the cached GEMM-layout copies made by `prepared_weight`: if a frozen parameter has changed
    since (load_state_dict / resume), re-derive its copies INTO THE SAME BUFFERS.  Returns the number of refreshed copies."""
    n = 0
    for w in params:
        f32 = w.__dict__.get("_t2v_f32")          # cached fp32 copy of a frozen non-fp32 vector (functional._f32)
        if f32 is not None:
            tag = (w.data_ptr(), w._version, tuple(w.shape), w.dtype)
            if f32[0] != tag and f32[0][2] == tag[2]:
                f32[1].copy_(w.detach().float())
                w.__dict__["_t2v_f32"] = (tag, f32[1])
                n += 1
        cache = w.__dict__.get("_t2v_prep")
        if not cache:
            continue
        tag = (w.data_ptr(), w._version, tuple(w.shape), w.dtype)
        for kind, hit in list(cache.items()):
            if isinstance(kind, tuple):          # ("group", fwd|bwd): concatenated copy of a projection group (cached on its first member)
                gtag, buf, ws = hit
                ntag = tuple((m.data_ptr(), m._version, tuple(m.shape)) for m in ws)
                if ntag != gtag:
                    if [t[2] for t in ntag] != [t[2] for t in gtag] or buf.device != w.device:
                        raise RuntimeError("t2v_amd: a frozen parameter changed shape/device under a captured step; re-capture")
                    buf.copy_(torch.cat([_prep_compute(m, kind[1], None) for m in ws], dim=0 if kind[1] == "fwd" else 1))
                    cache[kind] = (ntag, buf, ws)
                    n += 1
                continue
            old_tag, buf = hit
            if old_tag != tag:
                if old_tag[2] != tag[2] or buf.device != w.device:
                    raise RuntimeError("t2v_amd: a frozen parameter changed shape/device under a captured step; re-capture")
                buf.copy_(_prep_compute(w, kind, None))
                cache[kind] = (tag, buf)
                n += 1
    return n


def prepared_weight(w, kind):
    """bf16 GEMM-layout copy of a parameter.  Frozen parameters cache it ON THE PARAMETER OBJECT (validated against
    storage address + version, so a re-homed / updated / re-allocated tensor can never hit a stale copy); trainable
    ones are re-derived every call so optimizer updates (and graph replays) always see fresh values."""
    if w.requires_grad or not isinstance(w, torch.nn.Parameter):
        return _prep_compute(w, kind, None)
    cache = w.__dict__.setdefault("_t2v_prep", {})
    hit = cache.get(kind)
    tag = (w.data_ptr(), w._version, tuple(w.shape), w.dtype)
    if hit is None or hit[0] != tag:
        hit = (tag, _prep_compute(w, kind, None))
        cache[kind] = hit
    return hit[1]


def _f32(t):
    """fp32 view / copy of a bias-like vector.  A FROZEN Parameter stored in another dtype (the bf16 CLIP text tower: 10 vectors per
    layer) caches its copy on the Parameter object, validated like the prepared weights — a `.float()` per call was 233 cast
    launches per step."""
    if t is None:
        return None
    if t.dtype == torch.float32:
        return t.detach()
    if isinstance(t, torch.nn.Parameter) and not t.requires_grad:
        tag = (t.data_ptr(), t._version, tuple(t.shape), t.dtype)
        hit = t.__dict__.get("_t2v_f32")
        if hit is None or hit[0] != tag:
            hit = (tag, t.detach().float())
            t.__dict__["_t2v_f32"] = hit
        return hit[1]
    return t.detach().float()


def _pad_vec(v, n):
    if v is None or v.shape[0] == n:
        return v
    out = torch.zeros(n, dtype=v.dtype, device=v.device)
    out[: v.shape[0]] = v
    return out


# --------------------------------------------------------------------------- raw launches
def launch_gemm(cs=None, **kw):
    """Launch one GEMM-family problem.  `cs` asks for the GroupNorm column statistics of the stored output to ride in the
    epilogue (T2VGemm.colsum): {"mode": 1} or {"mode": 2, "x", "ldx", "sums", "gamma", "beta", "eps", "G", "silu",
    "domain_rows"}.  Returns (buffer, tile_rows) when the kernel the library selects can emit them, else None (the caller then
    runs the separate statistics kernel)."""
    g = make_gemm(**kw)
    out = None
    if cs is not None and _cs_enabled:
        bm = int(nv.lib().t2v_gemm_colsum_rows(C.byref(g)))
        if bm > 0 and (cs["mode"] == 1 or cs["domain_rows"] % bm == 0):
            nb = kw["n_split"] if kw.get("n_split", 0) > 0 else kw["N"]
            dev = torch.device("cuda", torch.cuda.current_device())
            buf = torch.empty(4 + (-(-kw["M"] // bm)) * nb * 2, dtype=torch.float32, device=dev)
            g.colsum, g.cs_mode = buf.data_ptr(), cs["mode"]
            if cs["mode"] == 2:
                g.cs_domain_rows = cs["domain_rows"]
                g.cs_x, g.cs_ldx = cs["x"], cs["ldx"]
                g.cs_sums, g.cs_gamma, g.cs_beta = cs["sums"], cs["gamma"], cs["beta"]
                g.cs_eps, g.cs_G, g.cs_silu = cs["eps"], cs["G"], cs["silu"]
                g.cs_drop_p, g.cs_drop_seed = cs.get("drop_p", 0.0), cs.get("drop_seed", 0)
            out = (buf, bm, kw["M"], nb)
    nv.call("t2v_gemm", C.byref(g), nv.stream())
    return out


_cs_enabled = os.environ.get("T2V_GN_FUSE", "1") != "0"     # A/B switch: GroupNorm statistics in the GEMM epilogues
_cs_last = [None]        # column statistics of the producer launched last (picked up by the wrapper that called it)
_gn_last = [None]        # forward record of the GroupNorm evaluated last (attached to its output by the wrapper)
_bwd_cs = {}             # data_ptr of a backward-data result -> (buffer, tile rows, M, C): consumed by the GroupNorm backward


def _lr_ok(kw):
    """True if the library runs this descriptor with its rank-wide epilogue term (T2VGemm.lr_*: 8-wave kernels only)."""
    return bool(nv.lib().t2v_gemm_lr_ok(C.byref(make_gemm(**kw))))


def launch_gemm_pair(kw_a, kw_b):
    nv.call("t2v_gemm_pair", C.byref(make_gemm(**kw_a)), C.byref(make_gemm(**kw_b)), nv.stream())


def make_gemm(*, M, N, K, A, lda, B, ldb, D, ldd, a_mode=0, a_trans=0, b_trans=0, b_conv=0, geom=None, out_mode=0,
                bias=None, rowbias=None, ldrb=0, rows_per_rb=0, R=None, ldr=0, alpha=1.0, beta=1.0, act=0, batch=1,
                strideA=0, strideB=0, strideD=0, strideR=0, split_k=1, drop_p=0.0, drop_seed=0, B2=None, ldb2=0, n_split=0,
                D2=None, ldd2=0, b_tapflip=0, b2_k0=0, b2_klen=0, use_ws=True, lr=None):
    g = Gemm = nv.Gemm()
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_mode, g.a_trans = A, lda, a_mode, a_trans
    g.B, g.ldb, g.b_trans, g.b_conv = B, ldb, b_trans, b_conv
    if geom is not None:
        g.geom = geom
    g.D, g.ldd, g.out_mode = D, ldd, out_mode
    g.bias = bias
    g.rowbias, g.ldrb, g.rows_per_rb = rowbias, ldrb, rows_per_rb
    g.R, g.ldr = R, ldr
    g.alpha, g.beta, g.act = alpha, beta, act
    g.batch, g.strideA, g.strideB, g.strideD, g.strideR = batch, strideA, strideB, strideD, strideR
    g.split_k = split_k
    g.drop_p, g.drop_seed = drop_p, drop_seed
    g.B2, g.ldb2, g.n_split, g.D2, g.ldd2, g.b_tapflip = B2, ldb2, n_split, D2, ldd2, b_tapflip
    g.b2_k0, g.b2_klen = b2_k0, b2_klen
    if lr is not None:      # rank-wide epilogue term (T2VGemm.lr_*): the LoRA branch of a dropped wrapper inside the base launch
        g.lr_mode, g.lr_rp, g.lr_taps = lr["mode"], lr["rp"], lr.get("taps", 1)
        g.lr_a, g.lr_lda = lr.get("a"), lr.get("lda", 0)
        g.lr_b, g.lr_ldb = lr["b"], lr["ldb"]
        g.lr_scale, g.lr_drop_p, g.lr_drop_seed = lr.get("scale", 1.0), lr.get("drop_p", 0.0), lr.get("drop_seed", 0)
        g.lr_group_cols = lr.get("group_cols", 0)
        for i, sd in enumerate(lr.get("group_seeds", ())[:2]):
            g.lr_group_seed[i] = sd
        g.lr_plane = lr.get("plane")
    if use_ws and out_mode == nv.OUT_BF16 and not a_trans and not b_trans and batch <= 1:
        ws = _gemm_workspace()          # one scratch per (device, stream): stream order serialises its users
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return g


_gemm_ws = {}


def _gemm_workspace():
    """fp32 scratch for library-side split-K: one per (device, stream) — stream order serialises the users of one buffer, and the
    trainer runs GEMMs on two streams at once (the CLIP text tower and the parameter refresh beside the VAE encode)."""
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    buf = _gemm_ws.get(key)
    if buf is None:
        # 64 MB per launch stream (288 GB of HBM: the trainer's three streams hold 192 MB); zero-filled: the first 64 KB hold the
        # arrival counters of the in-launch split-K reduction (t2v_abi.h), which every launch leaves zero
        buf = _gemm_ws[key] = torch.zeros(16 << 20, dtype=torch.float32, device=f"cuda:{dev}")
    return buf


def check_gemm_workspaces():
    """Host-side read of the split-K give-up flag (gemm_w8.hip: a reducer that times out waiting for its writers stores 0xdead in
    word 16383 of the workspace and sums incomplete slabs).  Synchronises; raises if any workspace carries the flag and re-zeroes
    the 64 KB of arrival counters so that later launches start from a clean state.  Meant for validation loops and the end of a
    training run (`DenoiseTrainer.check_device_flags`), not for the inside of a step."""
    bad = []
    for key, buf in _gemm_ws.items():
        if int(buf[16383:16384].view(torch.int32).item()) == 0xdead:
            bad.append(key)
            buf[:16384].zero_()
    if bad:
        raise RuntimeError(f"t2v_gemm: an in-launch split-K reduction timed out on (device, stream) {bad}: the results of that "
                           "step are wrong; the arrival counters have been re-armed")


def gemm_workspace_flags_async():
    """Enqueue a copy of every GEMM workspace's give-up word into pinned host memory (each on the current stream, behind the work
    already queued there) and return (event, [(key, workspace, pinned word)]) for `raise_on_gemm_flags` — the non-stalling form of
    `check_gemm_workspaces` that `DenoiseTrainer` polls every few steps."""
    items = []
    for key, buf in list(_gemm_ws.items()):
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(buf[16383:16384].view(torch.int32), non_blocking=True)
        items.append((key, buf, host))
    ev = torch.cuda.Event()
    ev.record()
    return ev, items


def raise_on_gemm_flags(pending):
    ev, items = pending
    ev.synchronize()
    bad = [key for key, _, host in items if int(host[0]) == 0xdead]
    if bad:
        for key, buf, _ in items:
            if key in bad:
                buf[:16384].zero_()
        raise RuntimeError(f"t2v_gemm: an in-launch split-K reduction timed out on (device, stream) {bad}: the gradients of a recent "
                           "step are wrong (restore the last checkpoint); the arrival counters have been re-armed")


_KMAJOR_WGS = int(os.environ.get("T2V_KMAJOR_WGS", "800"))


def _split_k(tiles, kdim):
    """K splits of a K-major launch with `tiles` 64x64 output tiles: about `_KMAJOR_WGS` workgroups, at least 128 reduction rows per
    split.  1600 was the optimum of round 3's K loop (scripts/kmajor_probe.py); with the round-6 loop (loads back to back, two
    younger steps kept in flight) a workgroup streams faster and the fp32-atomic hand-over of every extra split costs more than
    it buys: 800 is the measured optimum of the C3 step (profiles/r06_c3_split_policy.txt: 156.8 / 152.2 / 154.6 / 155.0 / 158.8 ms
    at 1600 / 800 / 560 / 400 / 280)."""
    return int(max(1, min(64, _KMAJOR_WGS // max(1, tiles), kdim // 128)))


# --------------------------------------------------------------------------- Linear / Conv (implicit GEMM)
class _ConvLinear(torch.autograd.Function):
    """y = act(alpha * drop(x (*) W^T) + bias + rowbias[img]) + residual      (x (*) W = linear or sliding window)"""

    @staticmethod
    def forward(ctx, x, weight, bias, rowbias, residual, cfg, alpha, drop_p, drop_seed, colsum=False, gn=None):
        x = _mat(x, "x")
        ctx.gn = gn
        wq = prepared_weight(weight, "fwd")
        npad = wq.shape[0]
        cin_p = wq.shape[1] // cfg.taps()
        if x.shape[1] != cin_p:
            raise RuntimeError(f"t2v_amd: input width {x.shape[1]} != prepared weight channels {cin_p}")
        if cfg.kind == "linear":
            M = x.shape[0]
        else:
            M = cfg.nimg * cfg.Ho * cfg.Wo
            if x.shape[0] != cfg.nimg * cfg.H * cfg.W:
                raise RuntimeError(f"t2v_amd: conv input rows {x.shape[0]} != nimg*H*W {cfg.nimg * cfg.H * cfg.W}")
        y = torch.empty(M, npad, dtype=BF16, device=x.device)
        b32 = _pad_vec(_f32(bias), npad)
        rpr = 0
        if rowbias is not None:
            rowbias = _mat(rowbias, "rowbias")
            rpr = M // rowbias.shape[0]
        if residual is not None:
            residual = _mat(residual, "residual")
        _cs_last[0] = launch_gemm(cs={"mode": 1} if (colsum and drop_p == 0.0) else None,
                                  M=M, N=npad, K=wq.shape[1], A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=wq.shape[1],
                                  D=y.data_ptr(), ldd=npad, a_mode=nv.A_DENSE if cfg.kind == "linear" else nv.A_CONV,
                                  geom=None if cfg.kind == "linear" else cfg.fwd_geom(cin_p), bias=nv.ptr(b32),
                                  rowbias=nv.ptr(rowbias), ldrb=_ld(rowbias) if rowbias is not None else 0, rows_per_rb=rpr,
                                  R=nv.ptr(residual), ldr=_ld(residual) if residual is not None else 0, alpha=alpha, beta=1.0,
                                  drop_p=drop_p, drop_seed=drop_seed)
        ctx.cfg, ctx.alpha, ctx.drop = cfg, alpha, (drop_p, drop_seed)
        ctx.has = (bias is not None and bias.requires_grad, rowbias is not None, residual is not None)
        ctx.bias_n = bias.shape[0] if bias is not None else 0
        ctx.save_for_backward(x, weight, rowbias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, rowbias = ctx.saved_tensors
        cfg, alpha = ctx.cfg, ctx.alpha
        drop_p, drop_seed = ctx.drop
        bias_grad, has_rb, has_res = ctx.has
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        dy = _mat(dy, "dy")
        M, npad = dy.shape
        dres = dy if has_res else None
        drb = None
        if has_rb:
            drb = rowgroup_sum(dy, rowbias.shape[0], M // rowbias.shape[0])
        db = None
        if bias_grad:
            db = dy.sum(0, dtype=torch.float32)[: ctx.bias_n]
        g = dy
        if drop_p > 0.0:
            # dropout sits between the GEMM and bias/residual: re-apply the same mask to dy
            g = torch.empty_like(dy)
            launch_gemm_dropmask(dy, g, drop_p, drop_seed)
        dx = None
        if ctx.needs_input_grad[0]:
            wb = prepared_weight(weight, "bwd")        # [Cin_p, taps*Np]
            cin_p = wb.shape[0]
            if cfg.kind == "linear":
                dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
                _note_bwd_cs(dx, launch_gemm(cs=_gn_bwd_request(ctx.gn, M, cin_p), M=M, N=cin_p, K=npad, A=g.data_ptr(), lda=_ld(g),
                                             B=wb.data_ptr(), ldb=wb.shape[1], D=dx.data_ptr(), ldd=cin_p, alpha=alpha))
            else:
                Hv, Wv = cfg.H << cfg.up, cfg.W << cfg.up
                Mi = cfg.nimg * Hv * Wv
                dxv = torch.empty(Mi, cin_p, dtype=BF16, device=dy.device)
                _note_bwd_cs(dxv, launch_gemm(cs=None if cfg.up else _gn_bwd_request(ctx.gn, Mi, cin_p), M=Mi, N=cin_p, K=wb.shape[1],
                                              A=g.data_ptr(), lda=_ld(g), B=wb.data_ptr(), ldb=wb.shape[1], D=dxv.data_ptr(),
                                              ldd=cin_p, a_mode=nv.A_CONV, geom=cfg.bwd_geom(npad), alpha=alpha))
                if cfg.up:
                    dx = torch.empty(cfg.nimg * cfg.H * cfg.W, cin_p, dtype=BF16, device=dy.device)
                    nv.call("t2v_pool2x2_sum", dxv.data_ptr(), cin_p, dx.data_ptr(), cin_p, cfg.nimg, cfg.H, cfg.W, cin_p,
                            nv.stream())
                else:
                    dx = dxv
        dw = None
        if ctx.needs_input_grad[1]:
            cin_p = x.shape[1]
            kw = cfg.taps() * cin_p
            # A Linear weight's GEMM layout IS its parameter layout: with a flat-buffer optimiser attached (`weight.grad` is
            # a view of the fp32 gradient buffer, training.FlatAdamW) the launch accumulates straight into it and autograd gets
            # no tensor to add — the zeros / un-permute / AccumulateGrad kernels were ~25 us per weight at config C3.
            # (only for parameters a FlatAdamW owns: a caller of torch.autograd.grad on a plain module must get its tensor back)
            wg = weight.grad if (weight.is_leaf and weight.__dict__.get("_t2v_flat_grad")) else None
            direct = (cfg.kind == "linear" and wg is not None and wg.dtype == torch.float32 and wg.is_contiguous()
                      and tuple(wg.shape) == (npad, kw) and not torch.is_grad_enabled())
            dwp = wg if direct else torch.zeros(npad, kw, dtype=torch.float32, device=dy.device)
            tiles = ((npad + 63) // 64) * ((kw + 63) // 64)
            launch_gemm(M=npad, N=kw, K=M, A=g.data_ptr(), lda=_ld(g), a_trans=1, B=x.data_ptr(), ldb=_ld(x), b_trans=1,
                        b_conv=0 if cfg.kind == "linear" else 1,
                        geom=None if cfg.kind == "linear" else cfg.fwd_geom(cin_p), D=dwp.data_ptr(), ldd=kw,
                        out_mode=nv.OUT_F32_ATOMIC, alpha=alpha, split_k=_split_k(tiles, M))
            dw = None if direct else _unprep_weight_grad(dwp, weight, cfg)
        return dx, dw, db, drb, dres, None, None, None, None, None, None


def rowgroup_sum(x, groups, rows_per_group):
    """[groups, cols] bf16 = sum over the `rows_per_group` consecutive rows of every group of x [groups * rows_per_group, cols]
    (fp32 accumulation): the gradient of a per-video row-bias (time embedding) or of keys / values shared by a video's frames."""
    x = _mat(x, "x")
    cols = x.shape[1]
    y = torch.empty(groups, cols, dtype=BF16, device=x.device)
    sp = int(nv.lib().t2v_rowgroup_splits(groups, rows_per_group, cols))
    ws = torch.empty(groups * sp * cols, dtype=torch.float32, device=x.device) if sp > 1 else None
    nv.call("t2v_rowgroup_sum", x.data_ptr(), _ld(x), y.data_ptr(), cols, groups, rows_per_group, cols, nv.ptr(ws), nv.stream())
    return y


def launch_gemm_dropmask(dy, out, drop_p, drop_seed):
    """Re-apply the GEMM epilogue's dropout mask (index = row * N + col) to the incoming gradient."""
    nv.call("t2v_dropout_mask", dy.data_ptr(), _ld(dy), out.data_ptr(), _ld(out), dy.shape[0], dy.shape[1], drop_p,
            drop_seed, nv.stream())


def _take_gn(x):
    """The forward record of the GroupNorm that produced `x`, for a layer that consumes x — and one more registered consumer.
    Only a norm output with exactly ONE consumer may get its backward sums from that consumer's backward-data launch: with
    several (the VAE attention feeds to_q / to_k / to_v) autograd adds their gradients in place into the first one's buffer, and
    sums taken from that one launch would describe a partial gradient."""
    rec = getattr(x, "_t2v_gn", None)
    if rec is not None:
        rec["uses"] = rec.get("uses", 0) + 1
    return rec


def _gn_bwd_request(gn, M, C_):
    """Column-statistics request for a backward-data launch whose result is the gradient of a GroupNorm output `gn` recorded:
    the two sums of t2v_gn_bwd_stats ride in the epilogue (mode 2).  None when the record does not fit this launch."""
    if gn is None or gn.get("uses", 1) != 1 or gn["x"].shape != (M, C_) or gn["want_pg"]:
        return None
    # (a norm that drops behind its SiLU — TemporalConvLayer — hands its mask over: cs_drop_p / cs_drop_seed, round 4)
    return {"mode": 2, "x": gn["x"].data_ptr(), "ldx": _ld(gn["x"]), "sums": gn["sums"].data_ptr(), "gamma": gn["g32"].data_ptr(),
            "beta": gn["b32"].data_ptr(), "eps": gn["eps"], "G": gn["G"], "silu": gn["silu"], "domain_rows": gn["rpd"],
            "drop_p": gn["drop_p"], "drop_seed": gn["drop_seed"]}


def _note_bwd_cs(dx, info):
    if info is not None:
        _bwd_cs[dx.data_ptr()] = info
        _join_at_end_of_backward()           # entries nobody consumed must not outlive the pass (a later tensor could land on the address)


class _Dropout(torch.autograd.Function):
    """nn.Dropout on a token matrix with the counter-based mask of the kernels (index = row * cols + col): the same call with
    the same seed re-applies the mask to the gradient."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _mat(x, "x")
        y = torch.empty(x.shape, dtype=BF16, device=x.device)
        nv.call("t2v_dropout_mask", x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), x.shape[0], x.shape[1], p, seed, nv.stream())
        ctx.args = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _mat(dy if dy.stride(1) == 1 else dy.contiguous(), "dy")
        dx = torch.empty(dy.shape, dtype=BF16, device=dy.device)
        launch_gemm_dropmask(dy, dx, *ctx.args)
        return dx, None, None


def dropout(x, p, seed):
    return _Dropout.apply(x, float(p), int(seed)) if p > 0.0 else x


def _unprep_weight_grad(dwp, weight, cfg):
    """[Np, taps*Cin_p] fp32 (prepared order) -> gradient in the parameter's own layout/dtype."""
    n = weight.shape[0]
    cin = weight.shape[1]
    taps = cfg.taps()
    d3 = dwp.view(dwp.shape[0], taps, -1)[:n, :, :cin]
    if weight.dim() == 2:
        gw = d3.reshape(n, cin)
    elif weight.dim() == 4:
        gw = d3.view(n, cfg.KH, cfg.KW, cin).permute(0, 3, 1, 2)
    else:
        gw = d3.permute(0, 2, 1).reshape(n, cin, taps, 1, 1)
    return gw.to(weight.dtype).contiguous()


# Factor-gradient launches (dU, dD) feed only the flat gradient buffer, i.e. they are off the critical path of backward.  By
# default their descriptors are queued and run in batches (further down).  With T2V_WGRAD_BATCH=0 they are per-layer launches
# on a side stream (forked after dt, joined once before the optimizer; a parallel branch of a captured graph).
_side = {"stream": None, "refs": [], "enabled": os.environ.get("T2V_WGRAD_STREAM", "1") != "0", "cb": False}


def _side_stream():
    if _side["stream"] is None:
        _side["stream"] = torch.cuda.Stream()
    return _side["stream"]


def join_side_stream():
    """Make the current stream wait for every factor-gradient launch issued so far; release their operand references."""
    flush_wgrads()
    if _side["stream"] is not None and _side["refs"]:
        torch.cuda.current_stream().wait_stream(_side["stream"])
    _side["refs"].clear()


# ---- factor-gradient batching: the t2v_lora_wgrad descriptors of a backward pass are queued and run as a few
# t2v_lora_wgrad_batch launches instead of one ~15 us kernel per layer.  A batch is flushed IN STREAM ORDER on the launch
# stream when it holds T2V_WGRAD_BATCH_LAYERS layers or T2V_WGRAD_BATCH_MB of operands (the operands are kept alive only until
# their batch is enqueued — stream order makes the later reuse of their memory safe — so the extra residency is bounded: at the
# C5 grid, holding every layer's x and dy to the end of the backward was +130 GB), and at the end of the backward.
# T2V_WGRAD_BATCH=0 restores the per-layer launches (on the side stream).
_wq = {"enabled": os.environ.get("T2V_WGRAD_BATCH", "1") != "0", "descs": [], "keep": [], "bytes": 0, "pool": [], "next": 0,
       "captured": [], "reserve": [], "flush_at": int(os.environ.get("T2V_WGRAD_BATCH_LAYERS", "96")),
       "flush_bytes": int(os.environ.get("T2V_WGRAD_BATCH_MB", "4096")) << 20}


def _wgrad_launch(w, keep):
    """Run (or queue) the factor gradients of one layer; `keep` = every tensor the descriptor points into."""
    if not _wq["enabled"]:
        nv.call("t2v_lora_wgrad", C.byref(w), nv.stream())
        return
    _wq["descs"].append(w)
    _wq["keep"].append(keep)
    _wq["bytes"] += int(w.rows) * (int(w.N) + int(w.C)) * 2
    if len(_wq["descs"]) >= _wq["flush_at"] or _wq["bytes"] >= _wq["flush_bytes"]:
        flush_wgrads()
    else:
        _join_at_end_of_backward()           # (outside a backward pass this flushes and joins at once)


_WQ_BYTES = 1 << 16       # one staging / table buffer (a batch of T2V_WGRAD_BATCH_LAYERS layers needs ~45 KB)


def _wgrad_staging(nbytes, device):
    """(pinned host staging, device table, event) for one batch.  Eager: a ring of reusable pairs, each guarded by the event of
    its last copy.  Inside a stream capture: a pair from a reserve that eager launches keep stocked (pinned memory cannot be
    allocated while a capture is open) and that is never reused — the graph re-reads the staging buffer at every replay."""
    if nbytes > _WQ_BYTES:
        raise RuntimeError(f"t2v_amd: factor-gradient batch of {nbytes} bytes exceeds the staging buffers (lower T2V_WGRAD_BATCH_LAYERS)")
    if torch.cuda.is_current_stream_capturing():
        if not _wq["reserve"]:
            raise RuntimeError("t2v_amd: no pinned staging buffer left for a captured factor-gradient batch; run one eager "
                               "train step before capturing (DenoiseTrainer.capture does)")
        host = _wq["reserve"].pop()
        pair = (host, torch.empty(_WQ_BYTES, dtype=torch.uint8, device=device), None)
        _wq["captured"].append(pair)
        return pair
    _wq["seen"] = _wq.get("seen", 0) + 1     # batches since the last end of a backward pass
    want = max(32, 2 * max(_wq["seen"], _wq.get("per_pass", 0)) + 8)
    while len(_wq["reserve"]) < want:        # stock for later captures: a captured step takes as many as an eager one flushes
        _wq["reserve"].append(torch.empty(_WQ_BYTES, dtype=torch.uint8, pin_memory=True))
    pool = _wq["pool"]
    if len(pool) < want // 2:                # a ring at least as long as one pass, so a reused slot is from an earlier step
        pool.append([torch.empty(_WQ_BYTES, dtype=torch.uint8, pin_memory=True),
                     torch.empty(_WQ_BYTES, dtype=torch.uint8, device=device), torch.cuda.Event()])
        return pool[-1]
    slot = pool[_wq["next"] % len(pool)]
    _wq["next"] += 1
    slot[2].synchronize()                    # its previous copy has been consumed (it is from an earlier step: no wait in practice)
    if slot[1].device != device:
        slot[1] = torch.empty(_WQ_BYTES, dtype=torch.uint8, device=device)
    return slot


def clear_bwd_colsums():
    """Forget backward column statistics nobody consumed (called between steps: their buffers must not outlive the pass)."""
    _bwd_cs.clear()


def drop_pending_wgrads():
    """Discard factor-gradient descriptors that were queued by a backward pass which never reached its end (an exception in the
    middle of it — the reference's loop swallows those, train.py:881-883): their operands may be gone.  Called at zero_grad()."""
    n = len(_wq["descs"])
    if n:
        _wq["descs"].clear()
        _wq["keep"].clear()
        _wq["bytes"] = 0
        _side["cb"] = False
        import warnings
        warnings.warn(f"t2v_amd: dropped {n} queued LoRA factor-gradient launches of an aborted backward pass")
    return n


def flush_wgrads():
    descs = _wq["descs"]
    n = len(descs)
    if not n:
        return
    arr = (nv.LoraWgrad * n)(*descs)
    descs.clear()
    nbytes = int(nv.lib().t2v_lora_wgrad_batch_bytes(n))
    host, dev, ev = _wgrad_staging(nbytes, torch.device("cuda", torch.cuda.current_device()))
    nv.call("t2v_lora_wgrad_batch", arr, n, host.data_ptr(), dev.data_ptr(), nbytes, nv.stream())
    if ev is not None:
        ev.record()
    _wq["keep"].clear()                      # enqueued on the launch stream: later reuse of this memory is ordered behind it
    _wq["bytes"] = 0


def _end_of_backward():
    _side["cb"] = False
    _bwd_cs.clear()
    join_side_stream()
    _wq["per_pass"] = max(_wq.get("per_pass", 0), _wq.get("seen", 0))
    _wq["seen"] = 0


def _fork_side(work, keep):
    """Run `work()` (factor-gradient launches) on the side stream, ordered after everything issued so far on the current
    stream.  The join is queued as an end-of-backward callback of the running autograd pass, so that whoever reads the
    gradients next (clip_grad_norm_, any optimizer, an all-reduce) sees them complete — also when the modules are driven by
    the reference's own train loop instead of DenoiseTrainer."""
    side = _side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work()
    _side["refs"].append(keep)               # keep operands alive until join_side_stream()
    _join_at_end_of_backward()


def _join_at_end_of_backward():
    if not _side["cb"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            _side["cb"] = True
        except RuntimeError:                 # not inside a backward pass (direct call of a Function's backward): join now
            join_side_stream()


def _lowrank_update(y, t, u, M, N, r, scale, drop_p=0.0, drop_seed=0):
    """y[M,N] += scale * drop(t[M,r] @ u[r,N]) (streaming rank-r update kernel; other ranks go through the GEMM)."""
    if r in (8, 16, 24, 32, 48, 64, 96):
        if drop_p > 0.0:
            nv.call("t2v_lowrank_update_drop", y.data_ptr(), _ld(y), t.data_ptr(), _ld(t), u.data_ptr(), _ld(u), M, N, r, scale,
                    drop_p, drop_seed, nv.stream())
        else:
            nv.call("t2v_lowrank_update", y.data_ptr(), _ld(y), t.data_ptr(), _ld(t), u.data_ptr(), _ld(u), M, N, r, scale,
                    nv.stream())
    elif drop_p > 0.0:
        raise RuntimeError(f"t2v_amd: LoRA dropout in the fused layer needs a padded rank in (8..96), got {r}")
    else:
        launch_gemm(M=M, N=N, K=r, A=t.data_ptr(), lda=_ld(t), B=u.data_ptr(), ldb=_ld(u), b_trans=1, D=y.data_ptr(), ldd=_ld(y),
                    R=y.data_ptr(), ldr=_ld(y), alpha=scale)


class _LoraLayer(torch.autograd.Function):
    """One LoRA-wrapped layer with the LoRA branch kept apart from the base weight:
      y = base(x) + scale * dropout(up(down(x)))        (utils/lora.py:57-62,134-139,211-216, identity selector)
    — the form needed when the wrapper's dropout is active (the reference's default train mode, dropout_p = 0.1:
    utils/lora.py:35,89), where the branch cannot be merged into the weight; also the T2V_LORA_MERGE=0 path.
      fwd : [y | t] = x (*) [W ; D]^T in ONE launch (the down projection rides as rank columns), y += s mask (t U^T)
      bwd : g = mask dy ; dt = g U ; dx = dy (*) W^T + s dt (*) D^T ; dU += s t^T g ; dD += s dt^T x
    With dropout off, g = dy and dt rides in the backward-data launch as rank columns."""

    @staticmethod
    def forward(ctx, x, w_base, b_base, down_w, up_w, rowbias, residual, cfg, e, scale, drop_p=0.0, drop_seed=0, colsum=False,
                gn=None):
        x = _mat(x, "x")
        ctx.gn = gn
        wq = prepared_weight(w_base, "fwd")
        npad, K = wq.shape
        cin_p = K // cfg.taps()
        if x.shape[1] != cin_p or npad != e.npad or cin_p != e.cin_p:
            raise RuntimeError("t2v_amd: LoRA bank entry does not match the layer")
        M = x.shape[0] if cfg.kind == "linear" else cfg.nimg * cfg.Ho * cfg.Wo
        y = torch.empty(M, npad, dtype=BF16, device=x.device)
        t = torch.empty(M, e.rp, dtype=BF16, device=x.device)
        b32 = _pad_vec(_f32(b_base), npad)
        rpr = 0
        if rowbias is not None:
            rowbias = _mat(rowbias, "rowbias")
            rpr = M // rowbias.shape[0]
        if residual is not None:
            residual = _mat(residual, "residual")
        conv = cfg.kind != "linear"
        # Epilogue form (round 4): every column tile of the base launch carries the rank rows of D, multiplies its own
        # t = x (*) D^T with U in the epilogue and adds s mask (t U^T) before the store — no pass over y, and the GroupNorm
        # statistics of the FINAL y can ride along.  Needs the step's transposed factor copies (lora_bank.PrepPlan) and a
        # descriptor the 8-wave kernels take.
        ctx.prep_ok = _lora_epi and getattr(e, "prep_scale", None) == scale and e.rp <= 32
        kw = None
        # keep-bit plane (round 6): the forward launch hashes the mask of its output anyway and leaves the bits (M * N / 8 bytes)
        # for the backward-data launch of a LINEAR wrapper, which then reads them in its K loop instead of hashing again
        plane = None
        if (ctx.prep_ok and _lora_plane and drop_p > 0.0 and not conv and npad % 64 == 0 and ctx.needs_input_grad[0] and
                M >= _LORA_EPI_MIN_ROWS_BWD and _dt_fuse_pays(M, cin_p, npad, e.rp)):
            plane = torch.empty(M * npad // 8, dtype=torch.uint8, device=x.device)
        if ctx.prep_ok and M >= _LORA_EPI_MIN_ROWS:
            kw = dict(M=M, N=npad, K=K, A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=K, D=y.data_ptr(), ldd=npad,
                      a_mode=nv.A_CONV if conv else nv.A_DENSE, geom=cfg.fwd_geom(cin_p) if conv else None,
                      bias=nv.ptr(b32), rowbias=nv.ptr(rowbias), ldrb=_ld(rowbias) if rowbias is not None else 0,
                      rows_per_rb=rpr, R=nv.ptr(residual), ldr=_ld(residual) if residual is not None else 0,
                      B2=e.down_w16.data_ptr(), ldb2=K, D2=t.data_ptr(), ldd2=e.rp,
                      lr=dict(mode=2, rp=e.rp, b=e.up_t16.data_ptr(), ldb=e.rk, scale=scale, drop_p=drop_p, drop_seed=drop_seed,
                              plane=nv.ptr(plane)))
            if not _lr_ok(kw):
                kw = None
        if kw is None:
            plane = None
        if kw is not None:
            _cs_last[0] = launch_gemm(cs={"mode": 1} if colsum else None, **kw)
        else:
            launch_gemm(M=M, N=npad + e.rp, K=K, A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=K, D=y.data_ptr(), ldd=npad,
                        a_mode=nv.A_CONV if conv else nv.A_DENSE, geom=cfg.fwd_geom(cin_p) if conv else None, bias=nv.ptr(b32),
                        rowbias=nv.ptr(rowbias), ldrb=_ld(rowbias) if rowbias is not None else 0, rows_per_rb=rpr,
                        R=nv.ptr(residual), ldr=_ld(residual) if residual is not None else 0,
                        B2=e.down_w16.data_ptr(), ldb2=K, n_split=npad, D2=t.data_ptr(), ldd2=e.rp)
            _lowrank_update(y, t, e.up_w16, M, npad, e.rp, scale, drop_p, drop_seed)     # y += s mask (t U^T)
        ctx.cfg, ctx.e, ctx.scale, ctx.drop = cfg, e, scale, (drop_p, drop_seed)
        ctx.has = (rowbias is not None, residual is not None)
        ctx.plane = plane
        ctx.save_for_backward(x, t, w_base, rowbias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, t, w_base, rowbias = ctx.saved_tensors
        cfg, e, scale = ctx.cfg, ctx.e, ctx.scale
        drop_p, drop_seed = ctx.drop
        has_rb, has_res = ctx.has
        dy = _mat(dy if dy.stride(1) == 1 else dy.contiguous(), "dy")
        M, npad = dy.shape
        conv = cfg.kind != "linear"
        dres = dy if has_res else None
        drb = None
        if has_rb:
            drb = rowgroup_sum(dy, rowbias.shape[0], M // rowbias.shape[0])
        cin_p = e.cin_p
        need_dx = ctx.needs_input_grad[0]
        # gradient of the LoRA branch output: g = mask dy / (1-p) with dropout on.  It is never materialised when the streaming
        # kernels apply: dt = g U comes from one masked pass over dy (t2v_lora_drop_dt) and the dU contraction regenerates the
        # mask itself (T2VLoraWgrad.drop_p) — round 3 wrote g (mask pass) and read it twice.
        g = dy
        stream_ok = e.rp <= 32 and x.shape[0] == M and (not conv or _wgrad_window_ok(cfg.fwd_geom(e.cin_p), M))
        fused_mask = drop_p > 0.0 and stream_ok and _drop_fuse
        if drop_p > 0.0 and not fused_mask:
            g = torch.empty_like(dy)
            launch_gemm_dropmask(dy, g, drop_p, drop_seed)
        dt = torch.empty(M, e.rp, dtype=BF16, device=dy.device)      # dt = g U (unscaled; `scale` is applied by its consumers)
        dx = None
        ride = drop_p == 0.0                         # dt can ride in the dx launch only when both read the same dy
        # Round 6 (T2VGemm.lr_mode 3): the backward-data launch of a LINEAR wrapper forms dt itself — rank fragments whose MFMAs
        # take masked A fragments (the mask hashed in the K loop) — and adds s dt D^T in its epilogue: ONE launch for dx and dt
        kw3 = None
        if (fused_mask and need_dx and not conv and ctx.prep_ok and x.shape[0] == M and M >= _LORA_EPI_MIN_ROWS_BWD and
                _dt_fuse_pays(M, cin_p, npad, e.rp)):
            wb = prepared_weight(w_base, "bwd")
            dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
            kw3 = dict(M=M, N=cin_p, K=wb.shape[1], A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(), ldb=wb.shape[1], D=dx.data_ptr(),
                       ldd=cin_p, B2=e.up_w16.data_ptr(), ldb2=_ld(e.up_w16), D2=dt.data_ptr(), ldd2=e.rp,
                       lr=dict(mode=3, rp=e.rp, b=e.down_t16.data_ptr(), ldb=_ld(e.down_t16), drop_p=drop_p, drop_seed=drop_seed,
                               plane=nv.ptr(ctx.plane)))
            if not _lr_ok(kw3):
                kw3, dx = None, None
        if kw3 is not None:
            _note_bwd_cs(dx, launch_gemm(cs=_gn_bwd_request(ctx.gn, M, cin_p), **kw3))
        elif fused_mask:
            nv.call("t2v_lora_drop_dt", dy.data_ptr(), _ld(dy), e.up_w16.data_ptr(), _ld(e.up_w16), dt.data_ptr(), e.rp, M, npad,
                    e.rp, drop_p, drop_seed, nv.stream())
        elif not ride:
            launch_gemm(M=M, N=e.rp, K=npad, A=g.data_ptr(), lda=_ld(g), B=e.up_w16.data_ptr(), ldb=_ld(e.up_w16),
                        D=dt.data_ptr(), ldd=e.rp)
        b2 = dict(B2=e.up_w16.data_ptr(), ldb2=_ld(e.up_w16), n_split=cin_p, D2=dt.data_ptr(), ldd2=e.rp) if ride else {}
        # Epilogue form: dx = dy (*) W^T + s dt (*) D^T in ONE launch — the rank-wide term is a few extra MFMAs per output
        # fragment on operands read straight from memory (dt, the flipped-tap transpose of s D), no pass over dx
        kw = None
        if kw3 is not None:
            pass
        elif (need_dx and not ride and ctx.prep_ok and x.shape[0] == M and M >= _LORA_EPI_MIN_ROWS_BWD and
                (not conv or (_wgrad_window_ok(cfg.fwd_geom(cin_p), M) and cfg.taps() in (1, 3, 9)))):
            wb = prepared_weight(w_base, "bwd")
            dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
            kw = dict(M=M, N=cin_p, K=wb.shape[1], A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(), ldb=wb.shape[1], D=dx.data_ptr(),
                      ldd=cin_p, a_mode=nv.A_CONV if conv else nv.A_DENSE, geom=cfg.bwd_geom(npad) if conv else None,
                      lr=dict(mode=1, rp=e.rp, taps=cfg.taps(), a=dt.data_ptr(), lda=e.rp, b=e.down_t16.data_ptr(),
                              ldb=_ld(e.down_t16)))
            if not _lr_ok(kw):
                kw, dx = None, None
        if kw3 is not None:
            pass
        elif kw is not None:
            _note_bwd_cs(dx, launch_gemm(cs=_gn_bwd_request(ctx.gn, M, cin_p), **kw))
        elif need_dx and not conv:
            # linear: [dx | dt] = dy [W^T | U] in ONE launch (dt rides as rp extra output columns)
            wb = prepared_weight(w_base, "bwd")
            dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
            launch_gemm(M=M, N=cin_p + (e.rp if ride else 0), K=npad, A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(),
                        ldb=wb.shape[1], D=dx.data_ptr(), ldd=cin_p, **b2)
            _lowrank_update(dx, dt, e.down_w16, M, cin_p, e.rp, scale)           # dx += s dt D
        elif need_dx and e.rp <= 32 and _wgrad_window_ok(cfg.fwd_geom(cin_p), M) and cfg.taps() in (3, 9):
            # stride-1 same-size conv: dt = dy U rides in the backward-data launch (rp extra output columns whose weights
            # exist only at the tap that gathers the row itself), then dx += s dt (*) D^T as a windowed rank update
            wb = prepared_weight(w_base, "bwd")
            bg = cfg.bwd_geom(npad)
            dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
            if ride:
                b2.update(b2_k0=(bg.py * bg.KW + bg.px) * npad, b2_klen=npad)
            launch_gemm(M=M, N=cin_p + (e.rp if ride else 0), K=wb.shape[1], A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(),
                        ldb=wb.shape[1], D=dx.data_ptr(), ldd=cin_p, a_mode=nv.A_CONV, geom=bg, **b2)
            fg = cfg.fwd_geom(cin_p)
            nv.call("t2v_lowrank_window_update", dx.data_ptr(), cin_p, dt.data_ptr(), e.rp, e.down_w16.data_ptr(),
                    cfg.taps() * cin_p, C.byref(fg), M, cin_p, e.rp, scale, nv.stream())
        else:
            if ride:
                launch_gemm(M=M, N=e.rp, K=npad, A=dy.data_ptr(), lda=_ld(dy), B=e.up_w16.data_ptr(), ldb=_ld(e.up_w16),
                            D=dt.data_ptr(), ldd=e.rp)
            if need_dx:
                wb = prepared_weight(w_base, "bwd")
                Hv, Wv = cfg.H << cfg.up, cfg.W << cfg.up
                Mi = cfg.nimg * Hv * Wv
                dxv = torch.empty(Mi, cin_p, dtype=BF16, device=dy.device)
                launch_gemm(M=Mi, N=cin_p, K=wb.shape[1], A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(), ldb=wb.shape[1],
                            D=dxv.data_ptr(), ldd=cin_p, a_mode=nv.A_CONV, geom=cfg.bwd_geom(npad))
                launch_gemm(M=Mi, N=cin_p, K=cfg.taps() * e.rp, A=dt.data_ptr(), lda=e.rp, B=e.down_w16.data_ptr(),
                            ldb=cfg.taps() * cin_p, b_trans=1, b_tapflip=1, D=dxv.data_ptr(), ldd=cin_p, R=dxv.data_ptr(),
                            ldr=cin_p, a_mode=nv.A_CONV, geom=cfg.bwd_geom(e.rp), alpha=scale)
                if cfg.up:
                    dx = torch.empty(cfg.nimg * cfg.H * cfg.W, cin_p, dtype=BF16, device=dy.device)
                    nv.call("t2v_pool2x2_sum", dxv.data_ptr(), cin_p, dx.data_ptr(), cin_p, cfg.nimg, cfg.H, cfg.W, cin_p,
                            nv.stream())
                else:
                    dx = dxv
        # factor gradients dU = s t^T g, dD = s dt^T x, accumulated in the flat fp32 gradient buffer (side stream)
        _lora_side_grads(x, g.data_ptr(), _ld(g), [g, dy, x, t, dt], cfg, e, scale, M, npad, cin_p, t=t, dt=dt,
                         drop=(drop_p, drop_seed) if fused_mask else None)
        return dx, None, None, None, None, drb, dres, None, None, None, None, None, None, None


_drop_fuse = os.environ.get("T2V_DROP_FUSE", "1") != "0"       # A/B switch: masks regenerated inside the backward kernels
_lora_epi = os.environ.get("T2V_LORA_EPI", "1") != "0"         # A/B switch: the dropped LoRA branch as an epilogue term of the base launch
_lora_plane = os.environ.get("T2V_LORA_PLANE", "1") != "0"      # A/B switch: keep-bit plane written by the forward launch, read by lr_mode 3
# dt of a dropped linear wrapper formed by its backward-data launch (lr_mode 3): "0" never, "1" where the estimate below says it pays,
# "2" wherever the kernels take the descriptor (A/B runs, tests)
_lora_dt_fuse = int(os.environ.get("T2V_LORA_DT_FUSE", "1") or 0)


def _dt_fuse_pays(M, N, K, rp, nmem=1):
    """Does forming dt inside the backward-data launch (lr_mode 3) beat the separate masked-dt launch?  Measured per signature on
    the C2 step (profiles/r06_dt_fuse_signatures.txt; in-step microseconds of the backward-data launch without -> with the rank
    fragments, against the 5.5 - 16 us of the dt launch it replaces): the launch gains one MFMA per A fragment and K step
    (+9 % at 320 base columns per tile, +20 % at 160) plus ~25 VALU instructions per masked fragment in ONE wave column, which the
    K loop hides only while it is short —
      level-0 projections   (32768, 320, K = 320)    +3.5 us  vs 11 us   fused
      level-1 projections   ( 8192, 640, K = 640)    +0   us  vs  6 us   fused
      level-0 q/k/v groups  (32768, 320, K = 960)    +5.6 us  vs 16 us   fused
      level-2 projections   ( 2048, 1280, K = 1280)  +5.7 us  vs 5.5 us  not fused (a wash)
      deeper groups / K >= 2560 / many column tiles  +11 .. +32 us vs 6 .. 16 us: not fused
    so: a layer of its own with K <= 640 and N <= 640, or a projection group at M >= 16384."""
    if _lora_dt_fuse >= 2:
        return True
    if _lora_dt_fuse <= 0:
        return False
    if nmem > 1:
        return M >= 16384 and K <= 1024
    return K <= 640 and N <= 640 and rp <= 32


_LORA_EPI_MIN_ROWS = int(os.environ.get("T2V_LORA_EPI_MIN_ROWS", "128"))
# the same for the backward-data launch (dx = dy (*) W^T + s dt (*) D^T as an epilogue term vs a plain launch + a rank update pass)
_LORA_EPI_MIN_ROWS_BWD = int(os.environ.get("T2V_LORA_EPI_MIN_ROWS_BWD", "128"))


def _lora_side_grads(x, dy_ptr, lddy, keep, cfg, e, scale, M, npad, cin_p, t=None, dt=None, drop=None):
    """Factor gradients of one merged layer on the side stream: dU += s t^T dy, dD += s dt^T x (streaming kernel; strided /
    resampled windows: two K-major GEMMs in one launch).  `t = x (*) D^T` normally rides in the forward launch and
    `dt = dy U` in the backward-data launch (rank columns of those launches); whichever is missing is formed here by a
    skinny rank-wide GEMM."""
    conv = cfg.kind != "linear"
    kw = cfg.taps() * cin_p

    def work():
        tt, dtt = t, dt
        g = cfg.fwd_geom(cin_p) if conv else None
        if tt is None:
            tt = torch.empty(M, e.rp, dtype=BF16, device=x.device)
            launch_gemm(M=M, N=e.rp, K=kw, A=x.data_ptr(), lda=_ld(x), B=e.down_w16.data_ptr(), ldb=kw, D=tt.data_ptr(),
                        ldd=e.rp, a_mode=nv.A_CONV if conv else nv.A_DENSE, geom=g, use_ws=False)
        if dtt is None:
            dtt = torch.empty(M, e.rp, dtype=BF16, device=x.device)
            launch_gemm(M=M, N=e.rp, K=npad, A=dy_ptr, lda=lddy, B=e.up_w16.data_ptr(), ldb=_ld(e.up_w16), D=dtt.data_ptr(),
                        ldd=e.rp, use_ws=False)
        if e.rp <= 32 and x.shape[0] == M and (not conv or _wgrad_window_ok(g, M)):
            w = nv.LoraWgrad()
            w.rows, w.rp, w.conv = M, e.rp, 1 if conv else 0
            w.t, w.ldt, w.dy, w.lddy, w.N = tt.data_ptr(), _ld(tt), dy_ptr, lddy, npad
            w.dU, w.lddu = e.up_g.data_ptr(), _ld(e.up_g)
            w.dt, w.lddt, w.x, w.ldx, w.C = dtt.data_ptr(), _ld(dtt), x.data_ptr(), _ld(x), cin_p
            w.dD, w.lddd = e.down_g.data_ptr(), kw
            if conv:
                w.geom = g
            w.alpha = scale
            if drop is not None:  # `dy_ptr` is the unmasked gradient: the dU contraction applies the branch's dropout mask itself
                w.drop_p, w.drop_seed = drop
            _wgrad_launch(w, (tt, dtt, x, keep))
        else:                     # strided / resampled windows: two K-major GEMMs in one launch
            assert drop is None
            launch_gemm_pair(
                dict(M=e.rp, N=npad, K=M, A=tt.data_ptr(), lda=_ld(tt), a_trans=1, B=dy_ptr, ldb=lddy, b_trans=1,
                     D=e.up_g.data_ptr(), ldd=_ld(e.up_g), out_mode=nv.OUT_F32_ATOMIC, alpha=scale,
                     split_k=_split_k((npad + 63) // 64, M)),
                dict(M=e.rp, N=kw, K=M, A=dtt.data_ptr(), lda=_ld(dtt), a_trans=1, B=x.data_ptr(), ldb=_ld(x), b_trans=1,
                     b_conv=1 if conv else 0, geom=g, D=e.down_g.data_ptr(), ldd=kw,
                     out_mode=nv.OUT_F32_ATOMIC, alpha=scale, split_k=_split_k((kw + 63) // 64, M)))
        keep.append((tt, dtt))       # the launches above are asynchronous: hold the temporaries until the join

    if _side["enabled"] and not _wq["enabled"]:
        _fork_side(work, keep)
    else:
        work()                   # batched factor gradients are queued and flushed in stream order (no side stream)


class _LoraMerged(torch.autograd.Function):
    """One LoRA-wrapped layer on its merged weight W_eff = W + s U D (lora_bank.MergePlan refreshes it once per step):
      fwd : y = x (*) W_eff^T  (+ bias, row-bias, residual)         — a plain N = C_out implicit GEMM
      bwd : dx = dy (*) W_eff ;  side stream: t = x (*) D^T, dt = dy U, dU += s t^T dy, dD += s dt^T x
    which is `base(x) + scale * up(down(x))` of utils/lora.py:57-62,134-139,211-216 (dropout off, identity selector) and
    its autograd, with no rank columns in the tile grid and no rank-update passes."""

    @staticmethod
    def forward(ctx, x, down_w, up_w, bias, rowbias, residual, cfg, e, scale, colsum=False, gn=None):
        x = _mat(x, "x")
        ctx.gn = gn
        wq = e.weff_fwd
        npad, K = e.npad, e.taps * e.cin_p
        if x.shape[1] != e.cin_p or cfg.taps() != e.taps:
            raise RuntimeError("t2v_amd: LoRA bank entry does not match the layer")
        conv = cfg.kind != "linear"
        M = cfg.nimg * cfg.Ho * cfg.Wo if conv else x.shape[0]
        if conv and x.shape[0] != cfg.nimg * cfg.H * cfg.W:
            raise RuntimeError(f"t2v_amd: conv input rows {x.shape[0]} != nimg*H*W {cfg.nimg * cfg.H * cfg.W}")
        y = torch.empty(M, npad, dtype=BF16, device=x.device)
        t = torch.empty(M, e.rp, dtype=BF16, device=x.device)       # t = x (*) D^T: rank columns of this launch, for dU only
        b32 = _pad_vec(_f32(bias), npad)
        rpr = 0
        if rowbias is not None:
            rowbias = _mat(rowbias, "rowbias")
            rpr = M // rowbias.shape[0]
        if residual is not None:
            residual = _mat(residual, "residual")
        _cs_last[0] = launch_gemm(cs={"mode": 1} if colsum else None,
                                  M=M, N=npad + e.rp, K=K, A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=_ld(wq), D=y.data_ptr(),
                                  ldd=npad, a_mode=nv.A_CONV if conv else nv.A_DENSE, geom=cfg.fwd_geom(e.cin_p) if conv else None,
                                  bias=nv.ptr(b32), rowbias=nv.ptr(rowbias), ldrb=_ld(rowbias) if rowbias is not None else 0,
                                  rows_per_rb=rpr, R=nv.ptr(residual), ldr=_ld(residual) if residual is not None else 0,
                                  B2=e.down_w16.data_ptr(), ldb2=K, n_split=npad, D2=t.data_ptr(), ldd2=e.rp)
        ctx.cfg, ctx.e, ctx.scale = cfg, e, scale
        ctx.has = (rowbias is not None, residual is not None)
        ctx.save_for_backward(x, rowbias, t)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, rowbias, t = ctx.saved_tensors
        cfg, e, scale = ctx.cfg, ctx.e, ctx.scale
        has_rb, has_res = ctx.has
        dy = _mat(dy if dy.stride(1) == 1 else dy.contiguous(), "dy")
        M, npad = dy.shape
        cin_p = e.cin_p
        dres = dy if has_res else None
        drb = None
        if has_rb:
            drb = rowgroup_sum(dy, rowbias.shape[0], M // rowbias.shape[0])
        dx, dt = None, None
        if ctx.needs_input_grad[0]:
            wb = e.weff_bwd                            # [Cin_p, taps*Np], flipped taps
            if cfg.kind == "linear":                   # [dx | dt] = dy [W_eff | U]: dt rides as rank columns
                dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
                dt = torch.empty(M, e.rp, dtype=BF16, device=dy.device)
                _note_bwd_cs(dx, launch_gemm(cs=_gn_bwd_request(ctx.gn, M, cin_p), M=M, N=cin_p + e.rp, K=npad, A=dy.data_ptr(),
                                             lda=_ld(dy), B=wb.data_ptr(), ldb=_ld(wb), D=dx.data_ptr(), ldd=cin_p,
                                             B2=e.up_w16.data_ptr(), ldb2=_ld(e.up_w16), n_split=cin_p, D2=dt.data_ptr(), ldd2=e.rp))
            elif e.rp <= 32 and _wgrad_window_ok(cfg.fwd_geom(cin_p), M) and cfg.taps() in (1, 3, 9):
                # stride-1 same-size window: the rank columns' weights exist only at the tap that gathers the row itself
                bg = cfg.bwd_geom(npad)
                dx = torch.empty(M, cin_p, dtype=BF16, device=dy.device)
                dt = torch.empty(M, e.rp, dtype=BF16, device=dy.device)
                _note_bwd_cs(dx, launch_gemm(cs=_gn_bwd_request(ctx.gn, M, cin_p), M=M, N=cin_p + e.rp, K=cfg.taps() * npad,
                                             A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(), ldb=_ld(wb), D=dx.data_ptr(), ldd=cin_p,
                                             a_mode=nv.A_CONV, geom=bg, B2=e.up_w16.data_ptr(), ldb2=_ld(e.up_w16), n_split=cin_p,
                                             D2=dt.data_ptr(), ldd2=e.rp, b2_k0=(bg.py * bg.KW + bg.px) * npad, b2_klen=npad))
            else:
                Hv, Wv = cfg.H << cfg.up, cfg.W << cfg.up
                Mi = cfg.nimg * Hv * Wv
                dxv = torch.empty(Mi, cin_p, dtype=BF16, device=dy.device)
                launch_gemm(M=Mi, N=cin_p, K=cfg.taps() * npad, A=dy.data_ptr(), lda=_ld(dy), B=wb.data_ptr(), ldb=_ld(wb),
                            D=dxv.data_ptr(), ldd=cin_p, a_mode=nv.A_CONV, geom=cfg.bwd_geom(npad))
                if cfg.up:
                    dx = torch.empty(cfg.nimg * cfg.H * cfg.W, cin_p, dtype=BF16, device=dy.device)
                    nv.call("t2v_pool2x2_sum", dxv.data_ptr(), cin_p, dx.data_ptr(), cin_p, cfg.nimg, cfg.H, cfg.W, cin_p,
                            nv.stream())
                else:
                    dx = dxv
        _lora_side_grads(x, dy.data_ptr(), _ld(dy), [dy, x, t, dt], cfg, e, scale, M, npad, cin_p, t=t, dt=dt)
        return dx, None, None, None, drb, dres, None, None, None, None, None


def lora_merged(x, bias, down_w, up_w, cfg, entry, scale, rowbias=None, residual=None, colsum=False):
    _cs_last[0] = None
    return _attach_cs(_LoraMerged.apply(x, down_w, up_w, bias, rowbias, residual, cfg, entry, float(scale), bool(colsum),
                                        _take_gn(x)))


class _LoraGroupMerged(torch.autograd.Function):
    """Projections sharing one input (q/k/v, or k/v of the text cross-attention) on their merged weights:
    [y_0 | .. | y_{n-1}] = x W_eff,cat^T ; backward dx = [dy_0 | ..] W_eff,cat ; factor gradients per member on the side
    stream from t_cat = x D_cat^T and dt_cat = dy_cat U_blk^T.  The factors are inputs only to keep the node alive when `x`
    carries no gradient (text states)."""

    @staticmethod
    def forward(ctx, x, g, scale, *factors):
        x = _mat(x, "x")
        wq = g.weff_fwd
        ncat, K = g.npad, g.cin_p
        if x.shape[1] != K:
            raise RuntimeError("t2v_amd: projection group does not match its layers")
        M = x.shape[0]
        y = torch.empty(M, ncat, dtype=BF16, device=x.device)
        t = torch.empty(M, g.rp, dtype=BF16, device=x.device)
        launch_gemm(M=M, N=ncat + g.rp, K=K, A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=_ld(wq), D=y.data_ptr(), ldd=ncat,
                    B2=g.down_w16.data_ptr(), ldb2=K, n_split=ncat, D2=t.data_ptr(), ldd2=g.rp)
        ctx.g, ctx.scale = g, scale
        ctx.save_for_backward(x, t)
        return tuple(y[:, i * g.npad_each:(i + 1) * g.npad_each] for i in range(g.n))

    @staticmethod
    def backward(ctx, *dys):
        x, t = ctx.saved_tensors
        g, scale = ctx.g, ctx.scale
        n, M = g.n, x.shape[0]
        if any(d is None for d in dys):
            dys = [d if d is not None else torch.zeros(M, g.npad_each, dtype=BF16, device=x.device) for d in dys]
        if _adjacent_columns(dys):
            dy_ptr, lddy = dys[0].data_ptr(), dys[0].stride(0)
            keep = [dys, x, t]
        else:
            dcat = torch.cat([_mat(d if d.stride(1) == 1 else d.contiguous(), "dy") for d in dys], dim=1)
            dy_ptr, lddy = dcat.data_ptr(), dcat.stride(0)
            keep = [dcat, x, t]
        cin_p, ncat = g.cin_p, g.npad
        wb = g.weff_bwd
        dx = torch.empty(M, cin_p, dtype=BF16, device=x.device)
        dt = torch.empty(M, g.rp, dtype=BF16, device=x.device)      # [dx | dt_cat] = dy_cat [W_eff,cat | U_blk]
        launch_gemm(M=M, N=cin_p + g.rp, K=ncat, A=dy_ptr, lda=lddy, B=wb.data_ptr(), ldb=_ld(wb), D=dx.data_ptr(), ldd=cin_p,
                    B2=g.up_w16.data_ptr(), ldb2=ncat, n_split=cin_p, D2=dt.data_ptr(), ldd2=g.rp)
        keep.append(dt)
        rpe, npe = g.rp_each, g.npad_each

        def work():
            for i in range(n):
                w = nv.LoraWgrad()
                w.rows, w.rp, w.conv = M, rpe, 0
                w.t, w.ldt = t.data_ptr() + i * rpe * 2, g.rp
                w.dy, w.lddy, w.N = dy_ptr + i * npe * 2, lddy, npe
                w.dU, w.lddu = g.up_g.data_ptr() + (i * rpe * ncat + i * npe) * 4, ncat
                w.dt, w.lddt = dt.data_ptr() + i * rpe * 2, g.rp
                w.x, w.ldx, w.C = x.data_ptr(), _ld(x), cin_p
                w.dD, w.lddd = g.down_g.data_ptr() + i * rpe * cin_p * 4, cin_p
                w.alpha = scale
                _wgrad_launch(w, (t, dt, x, keep))

        if _side["enabled"] and not _wq["enabled"]:
            _fork_side(work, keep)
        else:
            work()
        return (dx if ctx.needs_input_grad[0] else None, None, None) + (None,) * (2 * n)


def lora_group_merged(x, group, scale):
    factors = [w for m in group.mods for w in (m.lora_down.weight, m.lora_up.weight)]
    return _LoraGroupMerged.apply(x, group, float(scale), *factors)


def _group_weight(ws, kind):
    """Concatenated bf16 GEMM-layout copy of the (frozen) base weights of a projection group, cached on the first member:
    fwd [n*Np, K] (rows = outputs of all members), bwd [Kin_p, n*Np]."""
    w0 = ws[0]
    tag = tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in ws)
    cache = w0.__dict__.setdefault("_t2v_prep", {})
    hit = cache.get(("group", kind))
    if hit is None or hit[0] != tag:
        parts = [prepared_weight(w, kind) for w in ws]
        hit = (tag, torch.cat(parts, dim=0 if kind == "fwd" else 1).contiguous(), list(ws))
        cache[("group", kind)] = hit
    return hit[1]


def _adjacent_columns(ts):
    """True if the 2-D tensors are consecutive, equally wide column blocks of one row-major buffer."""
    t0 = ts[0]
    if any(t.dim() != 2 or t.dtype != t0.dtype or t.shape != t0.shape or t.stride() != t0.stride() for t in ts):
        return False
    w = t0.shape[1]
    if t0.stride(1) != 1 or t0.stride(0) < w * len(ts):
        return False
    return all(t.data_ptr() == t0.data_ptr() + i * w * t0.element_size() for i, t in enumerate(ts))


def column_blocks(rows, width, n, device):
    """n column blocks [rows, width] of one fresh [rows, n*width] bf16 buffer (what `_LoraGroup.backward` can consume
    without a concatenation copy)."""
    buf = torch.empty(rows, n * width, dtype=BF16, device=device)
    return [buf[:, i * width:(i + 1) * width] for i in range(n)]


class _LoraGroup(torch.autograd.Function):
    """Projections sharing one input (to_q/to_k/to_v, or to_k/to_v of the text cross-attention), each LoRA-wrapped, as ONE
    layer: [y_0 | .. | y_{n-1} | t] = x [W_0; ..; W_{n-1}; D_cat]^T, y += s t U_blk (U_blk block-diagonal, lora_bank.py);
    backward [dx | dt] = [dy_0 | .. ] [W_cat^T | U_blk^T], dx += s dt D_cat, factor gradients per member."""

    @staticmethod
    def forward(ctx, x, g, scale, *params):
        # params = the n frozen base weights followed by the 2n trainable factors.  The factors are inputs only so that the
        # node requires grad when `x` does not (text cross-attention: keys/values are projections of the text states, which
        # carry no gradient with a frozen text encoder) — their gradients are accumulated into the flat buffer, not returned.
        x = _mat(x, "x")
        n = g.n
        w_bases = params[:n]
        wq = _group_weight(w_bases, "fwd")
        ncat, K = wq.shape
        if x.shape[1] != K or ncat != g.npad or K != g.cin_p:
            raise RuntimeError("t2v_amd: projection group does not match its layers")
        M = x.shape[0]
        y = torch.empty(M, ncat, dtype=BF16, device=x.device)
        t = torch.empty(M, g.rp, dtype=BF16, device=x.device)
        launch_gemm(M=M, N=ncat + g.rp, K=K, A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=K, D=y.data_ptr(), ldd=ncat,
                    B2=g.down_w16.data_ptr(), ldb2=K, n_split=ncat, D2=t.data_ptr(), ldd2=g.rp)
        _lowrank_update(y, t, g.up_w16, M, ncat, g.rp, scale)
        ctx.g, ctx.scale = g, scale
        ctx.save_for_backward(x, t, *w_bases)
        return tuple(y[:, i * g.npad_each:(i + 1) * g.npad_each] for i in range(n))

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors              # ONE access: under torch.utils.checkpoint a second unpack of a recomputed tensor raises
        x, t = saved[:2]
        w_bases = saved[2:]
        g, scale = ctx.g, ctx.scale
        n, M = g.n, x.shape[0]
        need_dx = ctx.needs_input_grad[0]
        if any(d is None for d in dys):
            dys = [d if d is not None else torch.zeros(M, g.npad_each, dtype=BF16, device=x.device) for d in dys]
        if _adjacent_columns(dys):
            dy_ptr, lddy = dys[0].data_ptr(), dys[0].stride(0)
            keep = dys
        else:
            dcat = torch.cat([_mat(d if d.stride(1) == 1 else d.contiguous(), "dy") for d in dys], dim=1)
            dy_ptr, lddy = dcat.data_ptr(), dcat.stride(0)
            keep = (dcat,)
        wb = _group_weight(w_bases, "bwd")
        cin_p, ncat = g.cin_p, g.npad
        dx = torch.empty(M, cin_p, dtype=BF16, device=x.device)
        dt = torch.empty(M, g.rp, dtype=BF16, device=x.device)
        launch_gemm(M=M, N=cin_p + g.rp, K=ncat, A=dy_ptr, lda=lddy, B=wb.data_ptr(), ldb=ncat, D=dx.data_ptr(), ldd=cin_p,
                    B2=g.up_w16.data_ptr(), ldb2=ncat, n_split=cin_p, D2=dt.data_ptr(), ldd2=g.rp)
        _lowrank_update(dx, dt, g.down_w16, M, cin_p, g.rp, scale)
        rpe, npe = g.rp_each, g.npad_each

        def wgrads():
            for i in range(n):
                w = nv.LoraWgrad()
                w.rows, w.rp, w.conv = M, rpe, 0
                w.t, w.ldt = t.data_ptr() + i * rpe * 2, g.rp
                w.dy, w.lddy, w.N = dy_ptr + i * npe * 2, lddy, npe
                w.dU, w.lddu = g.up_g.data_ptr() + (i * rpe * ncat + i * npe) * 4, ncat
                w.dt, w.lddt = dt.data_ptr() + i * rpe * 2, g.rp
                w.x, w.ldx, w.C = x.data_ptr(), _ld(x), cin_p
                w.dD, w.lddd = g.down_g.data_ptr() + i * rpe * cin_p * 4, cin_p
                w.alpha = scale
                _wgrad_launch(w, (t, dt, x, keep))

        if _side["enabled"] and not _wq["enabled"]:
            _fork_side(wgrads, (keep, t, dt, x))
        else:
            wgrads()
        return (dx if need_dx else None, None, None) + (None,) * (3 * n)


class _LoraGroupDrop(torch.autograd.Function):
    """Projections sharing one input (to_q / to_k / to_v, or to_k / to_v of the text cross-attention) with their wrappers' dropout
    ACTIVE (the reference's default train mode), as ONE layer with the branches in the epilogue (T2VGemm.lr_mode 2 with
    lr_group_cols): [y_0 | .. | y_{n-1}] = x [W_0; ..]^T + s mask_i (t_i U_i^T), every member with its own seed and mask index —
    the masks it draws as a stand-alone layer; backward: dt_cat from one masked pass over [dy_0 | ..] (t2v_lora_drop_dt_group),
    dx = dy_cat W_cat + s dt_cat D_cat in one launch (lr_mode 1, 16 n ranks side by side), factor gradients per member."""

    @staticmethod
    def forward(ctx, x, g, scale, drop_p, seeds, *params):
        x = _mat(x, "x")
        n = g.n
        w_bases = params[:n]
        wq = _group_weight(w_bases, "fwd")
        ncat, K = wq.shape
        if x.shape[1] != K or ncat != g.npad or K != g.cin_p:
            raise RuntimeError("t2v_amd: projection group does not match its layers")
        M = x.shape[0]
        y = torch.empty(M, ncat, dtype=BF16, device=x.device)
        t = torch.empty(M, g.rp, dtype=BF16, device=x.device)
        plane = None                                   # keep bits of all members (member i at byte i * M * npad_each / 8), see _LoraLayer
        if (_lora_plane and drop_p > 0.0 and g.npad_each % 64 == 0 and g.rp <= 64 and ctx.needs_input_grad[0] and
                _dt_fuse_pays(M, K, ncat, g.rp, n)):
            plane = torch.empty(M * ncat // 8, dtype=torch.uint8, device=x.device)
        launch_gemm(M=M, N=ncat, K=K, A=x.data_ptr(), lda=_ld(x), B=wq.data_ptr(), ldb=K, D=y.data_ptr(), ldd=ncat,
                    B2=g.down_w16.data_ptr(), ldb2=K, D2=t.data_ptr(), ldd2=g.rp,
                    lr=dict(mode=2, rp=g.rp_each, b=g.up_t16.data_ptr(), ldb=g.rk, scale=scale, drop_p=drop_p, drop_seed=seeds[0],
                            group_cols=g.npad_each, group_seeds=seeds[1:], plane=nv.ptr(plane)))
        ctx.plane = plane
        ctx.g, ctx.scale, ctx.drop = g, scale, (drop_p, seeds)
        ctx.save_for_backward(x, t, *w_bases)
        return tuple(y[:, i * g.npad_each:(i + 1) * g.npad_each] for i in range(n))

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors              # ONE access: under torch.utils.checkpoint a second unpack of a recomputed tensor raises
        x, t = saved[:2]
        w_bases = saved[2:]
        g, scale = ctx.g, ctx.scale
        drop_p, seeds = ctx.drop
        n, M = g.n, x.shape[0]
        need_dx = ctx.needs_input_grad[0]
        if any(d is None for d in dys):
            dys = [d if d is not None else torch.zeros(M, g.npad_each, dtype=BF16, device=x.device) for d in dys]
        if _adjacent_columns(dys):
            dy_ptr, lddy = dys[0].data_ptr(), dys[0].stride(0)
            keep = dys
        else:
            dcat = torch.cat([_mat(d if d.stride(1) == 1 else d.contiguous(), "dy") for d in dys], dim=1)
            dy_ptr, lddy = dcat.data_ptr(), dcat.stride(0)
            keep = (dcat,)
        cin_p, ncat = g.cin_p, g.npad
        rpe, npe = g.rp_each, g.npad_each
        dt = torch.empty(M, g.rp, dtype=BF16, device=x.device)          # dt_i = (mask_i dy_i / (1-p)) U_i, side by side
        dx = None
        kw3 = None
        if need_dx and g.rp <= 64 and _dt_fuse_pays(M, cin_p, ncat, g.rp, n):
            # round 6 (lr_mode 3): dt_cat is formed BY the backward-data launch (rank fragments on masked A fragments; the members
            # partition K, each with its own seed and mask width) — no t2v_lora_drop_dt_group launch
            wb = _group_weight(w_bases, "bwd")
            dx = torch.empty(M, cin_p, dtype=BF16, device=x.device)
            kw3 = dict(M=M, N=cin_p, K=ncat, A=dy_ptr, lda=lddy, B=wb.data_ptr(), ldb=ncat, D=dx.data_ptr(), ldd=cin_p,
                       B2=g.up_w16.data_ptr(), ldb2=_ld(g.up_w16), D2=dt.data_ptr(), ldd2=g.rp,
                       lr=dict(mode=3, rp=g.rp, b=g.down_t16.data_ptr(), ldb=_ld(g.down_t16), drop_p=drop_p, drop_seed=seeds[0],
                               group_cols=npe if n > 1 else 0, group_seeds=seeds[1:], plane=nv.ptr(ctx.plane)))
            if not _lr_ok(kw3):
                kw3, dx = None, None
        if kw3 is not None:
            launch_gemm(**kw3)
        else:
            sd = (C.c_ulonglong * 3)(*(list(seeds) + [0] * (3 - n)))
            nv.call("t2v_lora_drop_dt_group", dy_ptr, lddy, g.up_w16.data_ptr(), _ld(g.up_w16), rpe * _ld(g.up_w16) + npe, dt.data_ptr(), g.rp,
                    M, npe, rpe, n, drop_p, sd, nv.stream())
        if need_dx and kw3 is None:
            wb = _group_weight(w_bases, "bwd")
            dx = torch.empty(M, cin_p, dtype=BF16, device=x.device)
            launch_gemm(M=M, N=cin_p, K=ncat, A=dy_ptr, lda=lddy, B=wb.data_ptr(), ldb=ncat, D=dx.data_ptr(), ldd=cin_p,
                        lr=dict(mode=1, rp=g.rp, taps=1, a=dt.data_ptr(), lda=g.rp, b=g.down_t16.data_ptr(), ldb=_ld(g.down_t16)))
        for i in range(n):
            w = nv.LoraWgrad()
            w.rows, w.rp, w.conv = M, rpe, 0
            w.t, w.ldt = t.data_ptr() + i * rpe * 2, g.rp
            w.dy, w.lddy, w.N = dy_ptr + i * npe * 2, lddy, npe
            w.dU, w.lddu = g.up_g.data_ptr() + (i * rpe * ncat + i * npe) * 4, ncat
            w.dt, w.lddt = dt.data_ptr() + i * rpe * 2, g.rp
            w.x, w.ldx, w.C = x.data_ptr(), _ld(x), cin_p
            w.dD, w.lddd = g.down_g.data_ptr() + i * rpe * cin_p * 4, cin_p
            w.alpha = scale
            w.drop_p, w.drop_seed = drop_p, seeds[i]
            _wgrad_launch(w, (t, dt, x, keep))
        return (dx, None, None, None, None) + (None,) * (3 * n)


def lora_group_drop_ok(x, group, scale):
    """True if the grouped epilogue form applies: transposed factor copies current, batched wgrad path, and BOTH descriptors it
    launches — forward (lr_mode 2 with lr_group_cols) and backward-data (lr_mode 1 with all members' ranks side by side) — are
    ones the library's LR kernels take (`t2v_gemm_lr_ok` on dummy-pointer descriptors of the real shapes: the answer is cached
    per shape, so a shape the kernels refuse falls back to per-member layers here instead of raising mid-backward)."""
    if not (_lora_epi and _drop_fuse and _wq["enabled"] and getattr(group, "prep_scale", None) == float(scale)):
        return False
    M, K = x.shape[0], group.cin_p
    if M < _LORA_EPI_MIN_ROWS:
        return False
    # (the descriptors are probed with the leading dimensions the launches use — the input's own row stride forward;
    #  backward the gradient arrives as column blocks of one [M, npad] buffer or is concatenated into one, `_LoraGroupDrop.backward`)
    ldx = _ld(x)
    key = (M, K, group.npad, group.npad_each, group.rp, group.rp_each, group.rk, _ld(group.down_t16), ldx)
    hit = _group_ok_cache.get(key)
    if hit is None:
        p16 = 16                                    # (aligned dummy addresses: t2v_gemm_lr_ok reads the descriptor only)
        fwd = dict(M=M, N=group.npad, K=K, A=p16, lda=ldx, B=p16, ldb=K, D=p16, ldd=group.npad, B2=p16, ldb2=K, D2=p16, ldd2=group.rp,
                   lr=dict(mode=2, rp=group.rp_each, b=p16, ldb=group.rk, scale=float(scale), drop_p=0.1, drop_seed=1,
                           group_cols=group.npad_each, group_seeds=(2, 3)[: max(0, group.n - 1)]))
        bwd = dict(M=M, N=K, K=group.npad, A=p16, lda=group.npad, B=p16, ldb=group.npad, D=p16, ldd=K,
                   lr=dict(mode=1, rp=group.rp, taps=1, a=p16, lda=group.rp, b=p16, ldb=_ld(group.down_t16)))
        try:
            hit = _lr_ok(fwd) and _lr_ok(bwd)
        except Exception as e:   # noqa: BLE001  (a descriptor make_gemm itself refuses)
            import warnings
            warnings.warn(f"t2v_amd: the grouped dropped-LoRA form was refused for M={M} K={K} N={group.npad} ({type(e).__name__}: {e}); "
                          f"these projections run member by member")
            hit = False
        _group_ok_cache[key] = hit
    return hit


_group_ok_cache = {}


def lora_group_drop(x, group, scale, w_bases, drop_p, seeds):
    factors = [w for m in group.mods for w in (m.lora_down.weight, m.lora_up.weight)]
    return _LoraGroupDrop.apply(x, group, float(scale), float(drop_p), tuple(int(s) for s in seeds), *w_bases, *factors)


def lora_group(x, group, scale, w_bases):
    factors = [w for m in group.mods for w in (m.lora_down.weight, m.lora_up.weight)]
    return _LoraGroup.apply(x, group, float(scale), *w_bases, *factors)


def _wgrad_window_ok(g, rows):
    """t2v_lora_wgrad handles stride-1 same-size windows of 1, 3 or 9 taps (everything but the 3 stride-2 downsamplers and
    the 3 nearest-upsampled convs of the UNet)."""
    return (g.sy == 1 and g.sx == 1 and g.tdiv == 1 and g.up == 0 and g.Hv == g.Ho and g.Wv == g.Wo
            and g.KH * g.KW in (1, 3, 9) and rows % (g.Hv * g.Wv) == 0)


def lora_layer(x, w_base, b_base, down_w, up_w, cfg, entry, scale, rowbias=None, residual=None, drop_p=0.0, drop_seed=0,
               colsum=False):
    _cs_last[0] = None
    return _attach_cs(_LoraLayer.apply(x, w_base, b_base, down_w, up_w, rowbias, residual, cfg, entry, float(scale), float(drop_p),
                                       int(drop_seed), bool(colsum), _take_gn(x)))


def _attach_cs(y):
    """Hand the column statistics of the launch that produced `y` to whoever normalises it next (group_norm reads the attribute)."""
    info, _cs_last[0] = _cs_last[0], None
    if info is not None:
        y._t2v_cs = info
    return y


def conv_linear(x, weight, bias=None, cfg=LINEAR, rowbias=None, residual=None, alpha=1.0, drop_p=0.0, drop_seed=0, colsum=False):
    _cs_last[0] = None
    return _attach_cs(_ConvLinear.apply(x, weight, bias, rowbias, residual, cfg, float(alpha), float(drop_p), int(drop_seed),
                                        bool(colsum), _take_gn(x)))


# --------------------------------------------------------------------------- GroupNorm (+SiLU, +dropout)
_gn_ws = {}


def _gn_workspace(ndomains, G, device):
    """Scratch for the fixed-order statistics reduction; one buffer per (device, stream), reused by every call on that stream
    (stream order serialises the stats -> finalize pairs that use it; the VAE encode of the NEXT step runs on the trainer's
    auxiliary stream beside the UNet's norms)."""
    need = int(nv.lib().t2v_gn_workspace_floats(ndomains, G))
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _gn_ws.get(key)
    if buf is None or buf.numel() < need:
        buf = _gn_ws[key] = torch.zeros(max(need, 1 << 16), dtype=torch.float32, device=device)
    return buf


class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, silu, ndomains, drop_p, drop_seed, passthrough=False, cs=None):
        x = _mat(x, "x")
        rows, Cc = x.shape
        rpd = rows // ndomains
        g32, b32 = _f32(gamma), _f32(beta)
        sums = torch.empty(ndomains * G * 2, dtype=torch.float32, device=x.device)
        s = nv.stream()
        if cs is not None and cs[2] == rows and cs[3] == Cc and rpd % cs[1] == 0 and _ld(x) == Cc:
            # the producing GEMM left per-tile column sums of x: a small finishing launch instead of a pass over x
            nv.call("t2v_gn_finish", cs[0].data_ptr(), ndomains, rpd, Cc, G, sums.data_ptr(), s)
        else:
            ws = _gn_workspace(ndomains, G, x.device)
            nv.call("t2v_gn_stats", x.data_ptr(), _ld(x), ndomains, rpd, Cc, G, sums.data_ptr(), ws.data_ptr(), s)
        y = torch.empty(rows, Cc, dtype=BF16, device=x.device)
        nv.call("t2v_gn_apply", x.data_ptr(), _ld(x), y.data_ptr(), Cc, ndomains, rpd, Cc, G, sums.data_ptr(),
                g32.data_ptr(), b32.data_ptr(), eps, int(silu), drop_p, drop_seed, s)
        ctx.args = (G, eps, int(silu), ndomains, rpd, drop_p, drop_seed)
        ctx.save_for_backward(x, gamma, beta, sums)
        # forward record for the layer that consumes y: its backward-data launch can then carry this norm's backward sums
        _gn_last[0] = dict(x=x, sums=sums, g32=g32, b32=b32, G=G, eps=eps, silu=int(silu), rpd=rpd, drop_p=drop_p, drop_seed=drop_seed,
                           want_pg=bool(gamma.requires_grad or beta.requires_grad))
        if passthrough:          # second output: x itself, for the residual use — its gradient is summed inside bwd_apply
            return y, x.detach()
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, sums = ctx.saved_tensors
        G, eps, silu, ndomains, rpd, drop_p, drop_seed = ctx.args
        if dy is None:
            return (dres,) + (None,) * 10
        dy = _mat(dy if dy.stride(1) == 1 else dy.contiguous(), "dy")
        if dres is not None:
            dres = _mat(dres if dres.stride(1) == 1 else dres.contiguous(), "dres")
        rows, Cc = x.shape
        g32, b32 = _f32(gamma), _f32(beta)
        s = nv.stream()
        want_pg = gamma.requires_grad or beta.requires_grad
        dgm = torch.zeros(Cc, dtype=torch.float32, device=x.device) if want_pg else None
        dbt = torch.zeros(Cc, dtype=torch.float32, device=x.device) if want_pg else None
        bsums = torch.empty(ndomains * G * 2, dtype=torch.float32, device=x.device)
        cs = _bwd_cs.pop(dy.data_ptr(), None)
        if (cs is not None and not want_pg and cs[2] == rows and cs[3] == Cc and rpd % cs[1] == 0 and _ld(dy) == Cc):
            nv.call("t2v_gn_finish", cs[0].data_ptr(), ndomains, rpd, Cc, G, bsums.data_ptr(), s)
        else:
            ws = _gn_workspace(ndomains, G, x.device)
            # (full finetune: per-workgroup partial rows of d gamma / d beta, reduced in fixed order — caller-owned scratch)
            pgw = torch.empty(int(nv.lib().t2v_gn_bwd_pg_floats(ndomains, rpd, Cc)), dtype=torch.float32, device=x.device) if want_pg else None
            nv.call("t2v_gn_bwd_stats", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), ndomains, rpd, Cc, G, sums.data_ptr(),
                    g32.data_ptr(), b32.data_ptr(), eps, silu, drop_p, drop_seed, bsums.data_ptr(), ws.data_ptr(), nv.ptr(dgm),
                    nv.ptr(dbt), nv.ptr(pgw), s)
        dx = torch.empty(rows, Cc, dtype=BF16, device=x.device)
        nv.call("t2v_gn_bwd_apply", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), Cc, ndomains, rpd, Cc, G,
                sums.data_ptr(), bsums.data_ptr(), g32.data_ptr(), b32.data_ptr(), eps, silu, drop_p, drop_seed,
                nv.ptr(dres), _ld(dres) if dres is not None else 0, s)
        return (dx, dgm.to(gamma.dtype) if want_pg else None, dbt.to(beta.dtype) if want_pg else None,
                None, None, None, None, None, None, None, None)


def _attach_gn(y):
    rec, _gn_last[0] = _gn_last[0], None
    if rec is not None and _cs_enabled:
        y._t2v_gn = rec
    return y


def group_norm(x, gamma, beta, G, eps, silu, ndomains, drop_p=0.0, drop_seed=0):
    return _attach_gn(_GroupNorm.apply(x, gamma, beta, G, float(eps), bool(silu), int(ndomains), float(drop_p), int(drop_seed),
                                       False, getattr(x, "_t2v_cs", None)))


def group_norm_res(x, gamma, beta, G, eps, silu, ndomains, drop_p=0.0, drop_seed=0):
    """(group_norm(x), x_res): use `x_res` wherever x itself is consumed again (the residual around the normalised branch);
    the two gradients are then summed inside the backward-apply kernel instead of by a separate add."""
    y, xr = _GroupNorm.apply(x, gamma, beta, G, float(eps), bool(silu), int(ndomains), float(drop_p), int(drop_seed), True,
                             getattr(x, "_t2v_cs", None))
    return _attach_gn(y), xr


# --------------------------------------------------------------------------- LayerNorm
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, passthrough=False):
        x = _mat(x, "x")
        rows, Cc = x.shape
        y = torch.empty(rows, Cc, dtype=BF16, device=x.device)
        stats = torch.empty(rows * 2, dtype=torch.float32, device=x.device)
        nv.call("t2v_layernorm_fwd", x.data_ptr(), _ld(x), y.data_ptr(), Cc, rows, Cc, _f32(gamma).data_ptr(),
                _f32(beta).data_ptr(), eps, stats.data_ptr(), nv.stream())
        ctx.save_for_backward(x, gamma, beta, stats)
        if passthrough:
            return y, x.detach()
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, stats = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None
        dy = _mat(dy if dy.stride(1) == 1 else dy.contiguous(), "dy")
        if dres is not None:
            dres = _mat(dres if dres.stride(1) == 1 else dres.contiguous(), "dres")
        rows, Cc = x.shape
        want_pg = gamma.requires_grad or beta.requires_grad
        dgm = torch.zeros(Cc, dtype=torch.float32, device=x.device) if want_pg else None
        dbt = torch.zeros(Cc, dtype=torch.float32, device=x.device) if want_pg else None
        dx = torch.empty(rows, Cc, dtype=BF16, device=x.device)
        pgw = torch.empty(int(nv.lib().t2v_layernorm_bwd_pg_floats(rows, Cc)), dtype=torch.float32, device=x.device) if want_pg else None
        nv.call("t2v_layernorm_bwd", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), Cc, rows, Cc,
                _f32(gamma).data_ptr(), stats.data_ptr(), nv.ptr(dgm), nv.ptr(dbt), nv.ptr(pgw), nv.ptr(dres),
                _ld(dres) if dres is not None else 0, nv.stream())
        return dx, dgm.to(gamma.dtype) if want_pg else None, dbt.to(beta.dtype) if want_pg else None, None, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, float(eps))


def layer_norm_res(x, gamma, beta, eps=1e-5):
    """(layer_norm(x), x_res) — see group_norm_res."""
    return _LayerNorm.apply(x, gamma, beta, float(eps), True)


# --------------------------------------------------------------------------- attention core
@dataclass(frozen=True)
class SeqLayout:
    """How (batch b, position s) of a token matrix is addressed, in ROWS:
    row = (b // bdiv) * hi + (b % bdiv) * lo + s * ss   (element offset = row * ld + head*64)."""
    nbatch: int
    S: int
    hi: int
    lo: int
    ss: int
    bdiv: int = 1


def _operand(t, lay):
    ld = _ld(t)
    return nv.AttnOperand(t.data_ptr(), lay.hi * ld, lay.lo * ld, lay.ss * ld, lay.bdiv)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, qlay, klay, scale, causal=False):
        q, k, v = _mat(q, "q"), _mat(k, "k"), _mat(v, "v")
        o = torch.empty(q.shape[0], heads * 64, dtype=BF16, device=q.device)
        lse = torch.empty(qlay.nbatch * heads * qlay.S, dtype=torch.float32, device=q.device)
        a = nv.Attn()
        a.nbatch, a.heads, a.Sq, a.Sk, a.scale = qlay.nbatch, heads, qlay.S, klay.S, scale
        a.q, a.k, a.v, a.o = _operand(q, qlay), _operand(k, klay), _operand(v, klay), _operand(o, qlay)
        a.lse = lse.data_ptr()
        a.causal = int(causal)
        nv.call("t2v_attn_fwd", C.byref(a), nv.stream())
        ctx.meta = (heads, qlay, klay, scale, int(causal))
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        heads, qlay, klay, scale, causal = ctx.meta
        do = _mat(do if do.stride(1) == 1 else do.contiguous(), "do")
        width = heads * 64
        fused_qkv = _adjacent_columns((q, k, v))
        if fused_qkv:       # q, k, v came out of one fused projection: lay the gradients out the same way (no concat copy)
            dq, dk_, dv_ = column_blocks(q.shape[0], width, 3, q.device)
        else:
            dq = torch.empty(q.shape[0], width, dtype=BF16, device=q.device)
        # text cross-attention shares one K/V across the frames of a video (klay.lo == 0, bdiv = frames):
        # dK/dV are produced per query batch and summed over the sharing group below
        shared = klay.bdiv > 1 and klay.lo == 0
        if shared:
            dk = torch.empty(qlay.nbatch * klay.S, width, dtype=BF16, device=q.device)
            dv = torch.empty_like(dk)
            dklay = SeqLayout(qlay.nbatch, klay.S, klay.S, 0, 1, 1)
        else:
            if fused_qkv:
                dk, dv = dk_, dv_
            elif _adjacent_columns((k, v)):
                dk, dv = column_blocks(k.shape[0], width, 2, q.device)
            else:
                dk = torch.empty(k.shape[0], width, dtype=BF16, device=q.device)
                dv = torch.empty_like(dk)
            dklay = klay
        delta = torch.empty_like(lse)
        a = nv.Attn()
        a.nbatch, a.heads, a.Sq, a.Sk, a.scale = qlay.nbatch, heads, qlay.S, klay.S, scale
        a.q, a.k, a.v, a.o = _operand(q, qlay), _operand(k, klay), _operand(v, klay), _operand(o, qlay)
        a.lse = lse.data_ptr()
        a.d_o, a.dq, a.dk, a.dv = _operand(do, qlay), _operand(dq, qlay), _operand(dk, dklay), _operand(dv, dklay)
        a.delta = delta.data_ptr()
        a.causal = causal
        nv.call("t2v_attn_bwd", C.byref(a), nv.stream())
        if shared:
            nb_kv = qlay.nbatch // klay.bdiv
            # (sum over the frames that share one K / V: [nb_kv, bdiv, S*width] -> [nb_kv, S*width])
            dk = rowgroup_sum(dk.view(nb_kv * klay.bdiv, klay.S * width), nb_kv, klay.bdiv).view(nb_kv * klay.S, width)
            dv = rowgroup_sum(dv.view(nb_kv * klay.bdiv, klay.S * width), nb_kv, klay.bdiv).view(nb_kv * klay.S, width)
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, heads, qlay, klay, scale=0.125, causal=False):
    """q: [rows_q, heads*64]; k, v: [rows_kv, heads*64] (must be dense row-major); layouts describe batching.  `causal`: query i
    attends to keys 0..i (self-attention of the CLIP text tower)."""
    return _Attention.apply(q, k, v, int(heads), qlay, klay, float(scale), bool(causal))


# --------------------------------------------------------------------------- fused temporal unit (forward only)
_temporal_fused = os.environ.get("T2V_TEMPORAL_FUSED", "1") != "0"
_temporal_ablate = [0]          # measurement only (scripts/temporal_fused_probe.py): T2VTemporalFused.ablate of the next launches


_temporal_fused_maxc = int(os.environ.get("T2V_TEMPORAL_FUSED_MAXC", "512"))


def temporal_fused_ok(C_, frames, policy=True):
    """The library has a one-launch kernel for `LN -> q,k,v -> FxF softmax -> PV -> out-proj -> + residual` at this width / clip
    length (csrc/temporal_fused.hip), the switch T2V_TEMPORAL_FUSED is on and — `policy` — the measured dispatch rule says it wins:
    widths up to 512 (profiles/r06_temporal_fused_probe.txt: 49 vs 109 us at C = 320, 117 vs 168 us at C = 512 on the C2 grid; the
    C = 640 kernel holds LN(x) AND the attention output of all ten heads in registers, spills, and loses to the separate launches at
    both grids: 168 vs 79 us, 359 vs 284 us).  T2V_TEMPORAL_FUSED_MAXC moves the bound."""
    if not (_temporal_fused and bool(nv.lib().t2v_temporal_fused_ok(int(C_), int(frames)))):
        return False
    return (not policy) or int(C_) <= _temporal_fused_maxc


def temporal_fused_prepare_wo(wo):
    """The output-projection weight in the layout `t2v_temporal_fused_fwd` reads (include/t2v_abi.h): bf16 [C, C] with the INPUT index
    permuted inside every group of 16 — stored position 16 g + 8 a + 4 b + c holds input 16 g + 8 b + 4 a + c (a, b in {0, 1},
    c in 0..3): the eight head dims one lane of an O^T accumulator holds per k-step become one 16-byte chunk."""
    n, k = wo.shape
    return wo.detach().to(BF16).view(n, k // 16, 2, 2, 4).transpose(2, 3).reshape(n, k).contiguous()


def temporal_attention_fused(x, gamma, beta, eps, wqkv, wo, bo, batch, frames, hw, scale=0.125):
    """out = x + softmax_F((LN(x) Wq^T)(LN(x) Wk^T)^T scale)(LN(x) Wv^T) Wo^T + bo in ONE launch, nothing kept for a backward
    (include/t2v_abi.h `t2v_temporal_fused_fwd`).  x: [batch*frames*hw, C] bf16 token matrix, rows (b, f, pixel); wqkv: bf16 [3C, C]
    (rows of to_q, to_k, to_v), wo: bf16 [C, C] as `temporal_fused_prepare_wo` lays it out."""
    if torch.is_grad_enabled() and (x.requires_grad or wqkv.requires_grad or wo.requires_grad):
        raise RuntimeError("t2v_amd: temporal_attention_fused is forward-only; call it under torch.no_grad()")
    x = _mat(x, "x")
    Cc = wo.shape[0]
    if x.shape[0] != batch * frames * hw or x.shape[1] < Cc or tuple(wqkv.shape) != (3 * Cc, Cc) or tuple(wo.shape) != (Cc, Cc):
        raise RuntimeError("t2v_amd: temporal_attention_fused: operand shapes do not match")
    if wqkv.dtype != BF16 or wo.dtype != BF16 or not wqkv.is_contiguous() or not wo.is_contiguous():
        raise RuntimeError("t2v_amd: temporal_attention_fused wants contiguous bf16 weights")
    out = torch.empty(x.shape[0], Cc, dtype=BF16, device=x.device)
    d = nv.TemporalFused()
    d.x, d.ldx, d.out, d.ldo = x.data_ptr(), _ld(x), out.data_ptr(), Cc
    d.wqkv, d.wo = wqkv.data_ptr(), wo.data_ptr()
    g32, b32, bo32 = _f32(gamma), _f32(beta), _f32(bo)
    d.bo, d.gamma, d.beta = nv.ptr(bo32), g32.data_ptr(), b32.data_ptr()
    d.eps, d.scale = float(eps), float(scale)
    d.B, d.F, d.HW, d.C = int(batch), int(frames), int(hw), int(Cc)
    d.ablate = _temporal_ablate[0]
    nv.call("t2v_temporal_fused_fwd", C.byref(d), nv.stream())
    return out


# --------------------------------------------------------------------------- GEGLU / SiLU
class _Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _mat(x, "x")
        rows, two = x.shape
        inner = two // 2
        y = torch.empty(rows, inner, dtype=BF16, device=x.device)
        nv.call("t2v_geglu_fwd", x.data_ptr(), _ld(x), y.data_ptr(), inner, rows, inner, nv.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _mat(dy if dy.stride(1) == 1 else dy.contiguous(), "dy")
        rows, two = x.shape
        dx = torch.empty(rows, two, dtype=BF16, device=x.device)
        nv.call("t2v_geglu_bwd", x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), dx.data_ptr(), two, rows, two // 2,
                nv.stream())
        return dx


def geglu(x):
    return _Geglu.apply(x)


class _Silu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        nv.require_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        nv.call("t2v_silu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), nv.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        nv.call("t2v_silu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), nv.stream())
        return dx


class _Gelu(torch.autograd.Function):
    """GELU on a bf16 tensor: kind 0 exact erf (`gelu`), 1 `quick_gelu` — the MLP activation of the CLIP text tower."""

    @staticmethod
    def forward(ctx, x, kind):
        nv.require_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        nv.call("t2v_gelu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), kind, nv.stream())
        ctx.kind = kind
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        nv.call("t2v_gelu_bwd", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ctx.kind, nv.stream())
        return dx, None


def gelu(x, kind=0):
    if x.dtype != BF16 or x.numel() % 8:
        raise RuntimeError("t2v_amd: gelu expects a bf16 tensor whose size is a multiple of 8")
    return _Gelu.apply(x, int(kind))


def silu(x):
    if x.dtype != BF16:
        raise RuntimeError("t2v_amd: silu expects bf16")
    return _Silu.apply(x)


# --------------------------------------------------------------------------- concat (skip connections)
class _Concat(torch.autograd.Function):
    """torch.cat([a, b], dim=channels) of token matrices (models/unet_3d_blocks.py:764,861)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _mat(a, "a"), _mat(b, "b")
        rows, ca, cb = a.shape[0], a.shape[1], b.shape[1]
        y = torch.empty(rows, ca + cb, dtype=BF16, device=a.device)
        s = nv.stream()
        nv.call("t2v_copy2d", a.data_ptr(), _ld(a), y.data_ptr(), ca + cb, rows, ca, 0, s)
        nv.call("t2v_copy2d", b.data_ptr(), _ld(b), y.data_ptr() + 2 * ca, ca + cb, rows, cb, 0, s)
        ctx.split = (ca, cb)
        return y

    @staticmethod
    def backward(ctx, dy):
        ca, cb = ctx.split
        return dy[:, :ca], dy[:, ca:]


def concat(a, b):
    return _Concat.apply(a, b)


# --------------------------------------------------------------------------- 4-channel boundary
def planar_f32_to_tokens(x, width=8):
    """(n, C, rows) fp32 contiguous -> [n*rows, width] bf16 with zero padded channels (latent/pixel entry)."""
    nv.require_cuda(x)
    n, Cc, rows = x.shape
    x = x.contiguous().float()
    y = torch.zeros(n * rows, width, dtype=BF16, device=x.device)
    nv.call("t2v_f32_planar_to_bf16_cl", x.data_ptr(), y.data_ptr(), width, n, Cc, rows, nv.stream())
    return y
