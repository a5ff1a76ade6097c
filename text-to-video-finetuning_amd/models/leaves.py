"""Native leaf modules: same class names, attribute names and state-dict keys as the diffusers leaves the
reference imports (`models/unet_3d_blocks.py:18-20`, `models/unet_3d_condition.py:22-26`), but every forward
runs hand-written HIP kernels on channels-last bf16 token matrices (`t2v_amd.functional`).

Drop-in contract kept: leaf layers are plain `nn.Linear/nn.Conv2d/nn.Conv3d` children reachable via
`parent._modules[name]`, so `utils/lora.py:_find_modules_v2` + `inject_trainable_lora_extended` (class-name ancestor
search, exact-class gate, shared base Parameters) work unchanged.  Parents never call `child(x)` on the device path;
they call `run_layer(child, ...)`, which recognises a swapped-in LoRA wrapper (cloneofsimo: `.linear|.conv`,
`.lora_down`, `.lora_up`, `.dropout`, `.scale`, `.selector`; loralib style: `.lora_A`, `.lora_B`, `.scaling`)
and evaluates `base(x) + dropout(up(selector(down(x)))) * scale` (utils/lora.py:57-62,134-139,211-216) with the
same kernels.
"""
import math
from types import SimpleNamespace

import torch
from torch import nn

from .. import functional as F
from ..functional import ConvCfg, LINEAR, SeqLayout

BF16 = torch.bfloat16


class Tok:
    """Channels-last token matrix of an image batch: m [n*h*w, C] bf16 (rows ordered (n, y, x))."""
    __slots__ = ("m", "n", "h", "w")

    def __init__(self, m, n, h, w):
        self.m, self.n, self.h, self.w = m, n, h, w

    @property
    def C(self):
        return self.m.shape[1]

    @staticmethod
    def from_nchw(x, pad_to=8):
        if not x.is_cuda:
            raise RuntimeError("t2v_amd: the native modules run on a ROCm device only (tensor is on CPU); "
                               "the CPU restatement lives in oracle/ and is test infrastructure")
        n, c, h, w = x.shape
        m = x.permute(0, 2, 3, 1).reshape(n * h * w, c).to(BF16)
        cp = F.ceil8(c) if pad_to else c
        if cp != c:
            m = torch.nn.functional.pad(m, (0, cp - c))
        return Tok(m.contiguous(), n, h, w)

    def to_nchw(self, channels=None, dtype=torch.float32):
        c = channels or self.C
        return self.m[:, :c].reshape(self.n, self.h, self.w, c).permute(0, 3, 1, 2).to(dtype)


class _Out(SimpleNamespace):
    pass


_seed_state = {"base": 0x5EED, "ctr": 0, "step": 0, "fwd": -1}


def set_dropout_seed(seed):
    _seed_state["base"], _seed_state["ctr"], _seed_state["step"], _seed_state["fwd"] = int(seed), 0, 0, -1


def begin_forward():
    """A root forward with autograd on begins (UNet3DConditionModel.forward calls this): dropout sites draw a fresh mask per
    FORWARD, not only per optimisation step — the reference's two separate `unet(...)` calls of a step (train.py:814-834) and the
    micro-steps of a gradient-accumulation window each see their own masks, like nn.Dropout draws them.  The index counts the
    forwards since the last optimiser step (0 for the first one: a trainer's stacked two-pass forward keeps its seeds); a
    checkpoint recompute runs sub-module forwards only, so it repeats the masks of the forward it belongs to."""
    _seed_state["fwd"] += 1


def _next_seed():
    _seed_state["ctr"] += 1
    return (_seed_state["base"] * 1000003 + _seed_state["ctr"]) & 0xFFFFFFFFFFFF


def assign_dropout_names(root):
    """Give every module of `root` its qualified name as the key of its dropout seed (the trainers call this after LoRA
    injection).  A site's seed is then a function of (base seed, module name, optimisation step) instead of the order in which
    the forward happens to reach it: a recompute (gradient checkpointing) draws the same mask without rewinding a counter, and
    the CPU oracle can restate the masks site by site whatever order ITS forward visits them in (oracle/dropout.py)."""
    for name, m in root.named_modules():
        m.__dict__["_t2v_name"] = name


def advance_dropout_step():
    """Next optimisation step: new masks for plain eager loops (the trainer's captured step moves the DEVICE epoch instead)."""
    _seed_state["step"] += 1
    _seed_state["fwd"] = -1


def _seed_for(mod, suffix=""):
    name = mod.__dict__.get("_t2v_name")
    if name is None:
        return _next_seed()               # stand-alone module (kernel tests): counter protocol
    import zlib
    key = zlib.crc32((name + suffix).encode())
    return (_seed_state["base"] * 1000003 + key * 97 + _seed_state["step"] * 7919 + max(0, _seed_state["fwd"]) * 104729) & 0xFFFFFFFFFFFF


def _low_rank_product(B, A):
    """`lora_B @ lora_A` of a loralib / stable_lora layer (stable_lora/lora.py:119-126,190-197) on THIS library's GEMM: the rows of
    B are the activation operand, A^T the weight, so autograd reaches both factors through the same kernels as every other layer
    (dB = d(delta) A^T by the backward-data launch, dA by the K-major weight-gradient launch) — round 4 formed it with torch.matmul,
    i.e. a vendor GEMM on a path row.  bf16 operands, fp32 accumulation; the sum W + s delta is rounded to bf16 for the layer's own
    launch anyway."""
    r, rp = A.shape[0], F.ceil8(A.shape[0])
    Bb = B.to(BF16)
    if rp != r:
        Bb = torch.nn.functional.pad(Bb, (0, rp - r))
    return F.conv_linear(Bb.contiguous(), A.t(), None, LINEAR)[:, : A.shape[1]].float()


_fold_no_grad = __import__("os").environ.get("T2V_FOLD_NO_GRAD", "1") != "0"     # (0: base + down + up launches in no-grad calls too)


def _drop_p(mod):
    return float(mod.p) if (isinstance(mod, nn.Dropout) and mod.training) else 0.0


def _folded_weight(mod, base):
    """`W + scale * up @ down` of a cloneofsimo wrapper (Linear / Conv2d / Conv3d: utils/lora.py:57-62,134-139,211-216 with the
    Dropout inactive and no selector) as a FROZEN bf16 Parameter in the base weight's shape — what a no-grad forward (sampling,
    validation: train.py:895-958) multiplies by, one launch per layer instead of base + down + up.  Cached on the wrapper; the tag
    carries the tensors' version counters and, for factors that train, functional.weights_epoch (this library's optimiser kernels
    move no version counter).  The low-rank product runs on this library's GEMM (bf16 factors, fp32 accumulation), the sum in fp32."""
    w, u, d, sc = base.weight, mod.lora_up.weight, mod.lora_down.weight, float(mod.scale)
    tag = (w.data_ptr(), w._version, u.data_ptr(), u._version, d.data_ptr(), d._version, sc, str(w.device),
           F.weights_epoch[0] if (w.requires_grad or u.requires_grad or d.requires_grad) else -1)
    hit = mod.__dict__.get("_t2v_folded")
    if hit is None or hit[0] != tag:
        with torch.no_grad():
            r = d.shape[0]
            delta = _low_rank_product(u.detach().reshape(u.shape[0], r), d.detach().reshape(r, -1))
            wf = (w.detach().float() + sc * delta.view(w.shape)).to(BF16)
        hit = (tag, nn.Parameter(wf, requires_grad=False))
        mod.__dict__["_t2v_folded"] = hit
    return hit[1]


def run_layer(mod, x, cfg=LINEAR, rowbias=None, residual=None, colsum=False):
    """Evaluate a Linear/Conv leaf (or a LoRA wrapper swapped in for it) on a token matrix.  `colsum`: the output feeds a
    GroupNorm — let the GEMM epilogue leave that norm's statistics behind (functional.launch_gemm)."""
    if isinstance(mod, (nn.Linear, nn.Conv2d, nn.Conv3d)) and not hasattr(mod, "lora_A"):
        return F.conv_linear(x, mod.weight, mod.bias, cfg, rowbias, residual, colsum=colsum)
    base = getattr(mod, "linear", None)
    if base is None:
        base = getattr(mod, "conv", None)
    if base is not None and hasattr(mod, "lora_down") and hasattr(mod, "lora_up"):      # cloneofsimo wrapper
        entry = getattr(mod, "_t2v_bank", None)
        sel = getattr(mod, "selector", None)
        p = _drop_p(getattr(mod, "dropout", None))
        if (not torch.is_grad_enabled() and p == 0.0 and (sel is None or isinstance(sel, nn.Identity)) and _fold_no_grad
                and mod.lora_up.weight.shape[0] % 8 == 0 and base.weight.shape[1] % 8 == 0):
            # forward-only call with the branch's Dropout inactive: the folded weight, one launch
            return F.conv_linear(x, _folded_weight(mod, base), base.bias, cfg, rowbias, residual, colsum=colsum)
        if (entry is not None and not base.weight.requires_grad
                and (base.bias is None or not base.bias.requires_grad) and (sel is None or isinstance(sel, nn.Identity))
                and torch.is_grad_enabled()):
            if p == 0.0 and getattr(entry, "merge_scale", None) == float(mod.scale):   # merged weight W + s U D is current
                return F.lora_merged(x, base.bias, mod.lora_down.weight, mod.lora_up.weight, cfg, entry, float(mod.scale),
                                     rowbias, residual, colsum=colsum)
            if p == 0.0 or entry.rp in (8, 16, 24, 32, 48, 64, 96):
                # LoRA branch kept apart from the weight: active dropout (the reference's default train mode), or merge off
                return F.lora_layer(x, base.weight, base.bias, mod.lora_down.weight, mod.lora_up.weight, cfg, entry,
                                    float(mod.scale), rowbias, residual, drop_p=p, drop_seed=_seed_for(mod) if p > 0 else 0,
                                    colsum=colsum)
        y = F.conv_linear(x, base.weight, base.bias, cfg, rowbias, residual)
        t = F.conv_linear(x, mod.lora_down.weight, None, cfg)
        sel = getattr(mod, "selector", None)
        if sel is not None and not isinstance(sel, nn.Identity):
            t = F.conv_linear(t, sel.weight, None, LINEAR)
        p = _drop_p(getattr(mod, "dropout", None))
        return F.conv_linear(t, mod.lora_up.weight, None, LINEAR, None, y, alpha=float(mod.scale), drop_p=p,
                             drop_seed=_seed_for(mod) if p > 0 else 0)
    if hasattr(mod, "lora_A") and hasattr(mod, "lora_B"):                               # loralib / stable_lora style
        w = mod.weight
        merged = bool(getattr(mod, "merged", False))
        if mod.__dict__.get("_t2v_merged_seen", merged) != merged:
            # loralib folds / unfolds the delta with `weight.data +=` on train()/eval() (stable_lora/lora.py:112-117 upstream
            # loralib), which leaves the tensor's version counter untouched: drop the cached bf16 copies of the old values
            w.__dict__.pop("_t2v_prep", None)
        mod.__dict__["_t2v_merged_seen"] = merged
        p = _drop_p(getattr(mod, "lora_dropout", None))
        if p > 0.0 and w.dim() == 2 and getattr(mod, "r", 0) > 0 and not getattr(mod, "merged", False):
            # loralib Linear with an active lora_dropout: x W^T + (drop(x) A^T B^T) * scaling — the dropout sits on the INPUT of
            # the low-rank branch, so the branch cannot be folded into the weight (the conv flavours have no dropout in forward)
            y = F.conv_linear(x, w, mod.bias, cfg, rowbias, residual)
            t = F.conv_linear(F.dropout(x, p, _seed_for(mod, ".lora_dropout")), mod.lora_A, None, LINEAR)
            return F.conv_linear(t, mod.lora_B, None, LINEAR, None, y, alpha=float(mod.scaling))
        if getattr(mod, "r", 0) > 0 and not getattr(mod, "merged", False):
            delta = _low_rank_product(mod.lora_B, mod.lora_A)
            if w.dim() == 5:   # stable_lora Conv3d: view(out,in,k,k,1).mean(-2)  (stable_lora/lora.py:148-149,194)
                delta = delta.view(w.shape[0], w.shape[1], w.shape[2], w.shape[2], 1).mean(dim=-2, keepdim=True) \
                    .view(w.shape)
            else:
                delta = delta.view(w.shape)
            w = w + delta * mod.scaling
        return F.conv_linear(x, w, mod.bias, cfg, rowbias, residual)
    raise RuntimeError(f"t2v_amd: don't know how to run layer of type {type(mod).__name__} natively")


# --------------------------------------------------------------------------- time embedding
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        x = sample.to(BF16).contiguous()
        return run_layer(self.linear_2, F.silu(run_layer(self.linear_1, x)))


class Temb:
    """Per-video time embedding [B, 1280] with its SiLU cached (every ResnetBlock2D applies the same SiLU)."""

    def __init__(self, raw):
        self.raw = raw
        self._act = None

    @property
    def act(self):
        if self._act is None:
            self._act = F.silu(self.raw)
        return self._act


# --------------------------------------------------------------------------- ResnetBlock2D
class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, eps=1e-5, groups=32, dropout=0.0,
                 time_embedding_norm="default", non_linearity="silu", output_scale_factor=1.0, pre_norm=True):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        if self.output_scale_factor != 1.0:
            raise RuntimeError("t2v_amd: output_scale_factor != 1 is not used by the reference configs")
        cfg3 = ConvCfg.conv2d(x.n, x.h, x.w, 3, 1, 1)
        # xr: x for its second (shortcut / residual) use — its gradient is summed inside norm1's backward kernel
        a, xr = F.group_norm_res(x.m, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, True, x.n)
        rb = None
        if temb is not None and self.time_emb_proj is not None:
            rb = run_layer(self.time_emb_proj, temb.act)          # [B, Cout]; broadcast over the B's F*h*w rows
        h = run_layer(self.conv1, a, cfg3, rowbias=rb, colsum=True)
        a2 = F.group_norm(h, self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps, True, x.n,
                          _drop_p(self.dropout), _seed_for(self, ".dropout") if _drop_p(self.dropout) > 0 else 0)
        sc = xr
        if self.conv_shortcut is not None:
            sc = run_layer(self.conv_shortcut, xr, ConvCfg.conv2d(x.n, x.h, x.w, 1, 1, 0))
        out = run_layer(self.conv2, a2, cfg3, residual=sc, colsum=True)      # (feeds the next module's GroupNorm)
        return Tok(out, x.n, x.h, x.w)


# --------------------------------------------------------------------------- TemporalConvLayer
class TemporalConvLayer(nn.Module):
    def __init__(self, in_dim, out_dim=None, dropout=0.0):
        super().__init__()
        out_dim = out_dim or in_dim
        self.in_dim, self.out_dim = in_dim, out_dim
        self.conv1 = nn.Sequential(nn.GroupNorm(32, in_dim), nn.SiLU(),
                                   nn.Conv3d(in_dim, out_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv2 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv3 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv4 = nn.Sequential(nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, x, num_frames=1):
        B = x.n // num_frames
        cfg = ConvCfg.conv3d_t(B, num_frames, x.h * x.w)
        cur, xr = x.m, None
        seqs = (self.conv1, self.conv2, self.conv3, self.conv4)
        for i, seq in enumerate(seqs):
            gn, conv = seq[0], seq[-1]
            p = max((_drop_p(mm) for mm in seq), default=0.0)
            if i == 0:       # xr: the identity branch's use of x (gradient summed inside the first norm's backward kernel)
                a, xr = F.group_norm_res(cur, gn.weight, gn.bias, gn.num_groups, gn.eps, True, B, p,
                                          _seed_for(self, f".conv{i + 1}") if p > 0 else 0)
            else:
                a = F.group_norm(cur, gn.weight, gn.bias, gn.num_groups, gn.eps, True, B, p, _seed_for(self, f".conv{i + 1}") if p > 0 else 0)
            cur = run_layer(conv, a, cfg, residual=xr if i == 3 else None, colsum=True)
        return Tok(cur, x.n, x.h, x.w)


# --------------------------------------------------------------------------- attention / transformer blocks
class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False):
        super().__init__()
        if dim_head != 64:
            raise ValueError("t2v_amd: the native attention kernel is specialised for head_dim 64")
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])
        self.processor = None

    def set_processor(self, processor):   # train.py:138-139 — accepted; the native core is always used on device
        self.processor = processor

    def _fused_group(self, self_attention):
        """The projection group prepared by the trainer (lora_bank.py) if it can be used for this call: q,k,v for
        self-attention, k,v when keys/values come from another sequence; same conditions as run_layer's fused path."""
        g = self.__dict__.get("_t2v_group")
        if g is None or not torch.is_grad_enabled() or g.n != (3 if self_attention else 2):
            return None
        p0 = _drop_p(getattr(g.mods[0], "dropout", None))
        for m in g.mods:
            base = m.linear
            sel = getattr(m, "selector", None)
            # (dropout: all members off, or all on with one rate — the grouped epilogue form, functional.lora_group_drop)
            if (_drop_p(getattr(m, "dropout", None)) != p0 or base.weight.requires_grad or base.bias is not None
                    or not (sel is None or isinstance(sel, nn.Identity)) or float(m.scale) != float(g.mods[0].scale)
                    or getattr(m, "_t2v_bank", None) is None):
                return None
        return g

    def _no_grad_group(self, mods):
        """One frozen bf16 Parameter [sum N_i, K] = the rows of the members' effective weights (W, or W + s up down of a wrapper
        with inactive dropout), for a forward-only call: to_q | to_k | to_v (or to_k | to_v over the text states) as ONE launch —
        the grouped projections of the training forward (lora_bank.py) need a trainer; sampling has none.  None: a member has a
        bias or a flavour `_effective_weight` does not express."""
        parts = [_effective_weight(m) for m in mods]
        if any(p is None or p[2] is not None for p in parts):
            return None
        tag = tuple(p[0] for p in parts)
        hit = self.__dict__.get("_t2v_nograd_w", {}).get(len(mods))
        if hit is None or hit[0] != tag:
            ws = [p[1]() for p in parts]
            hit = (tag, nn.Parameter(torch.cat(ws, dim=0).contiguous(), requires_grad=False), [w.shape[0] for w in ws])
            self.__dict__.setdefault("_t2v_nograd_w", {})[len(mods)] = hit
        return hit

    def forward(self, x, qlay, ctx=None, klay=None, residual=None):
        src = x if ctx is None else ctx
        if not torch.is_grad_enabled() and _fold_no_grad:
            mods = (self.to_q, self.to_k, self.to_v) if ctx is None else (self.to_k, self.to_v)
            hit = self._no_grad_group(mods)
            if hit is not None and hit[1].device == src.device and hit[1].shape[1] == src.shape[1]:
                y = F.conv_linear(src, hit[1], None)
                cols, off = [], 0
                for n_ in hit[2]:
                    cols.append(y[:, off:off + n_])
                    off += n_
                q = cols[0] if ctx is None else run_layer(self.to_q, x)
                k, v = cols[-2], cols[-1]
                o = F.attention(q, k, v, self.heads, qlay, qlay if klay is None else klay, self.scale)
                return run_layer(self.to_out[0], o, residual=residual)
        g = self._fused_group(ctx is None)
        if g is not None:
            p0 = _drop_p(getattr(g.mods[0], "dropout", None))
            if p0 > 0.0 and not F.lora_group_drop_ok(src, g, g.mods[0].scale):
                g = None                 # dropping members without the grouped epilogue form: one by one (run_layer)
        merged = g is not None and getattr(g, "merge_scale", None) == float(g.mods[0].scale)

        def group(inp):
            if p0 > 0.0:
                return F.lora_group_drop(inp, g, g.mods[0].scale, [m.linear.weight for m in g.mods], p0, [_seed_for(m) for m in g.mods])
            if merged:
                return F.lora_group_merged(inp, g, g.mods[0].scale)
            return F.lora_group(inp, g, g.mods[0].scale, [m.linear.weight for m in g.mods])
        if g is not None and ctx is None:
            q, k, v = group(x)
        elif g is not None:
            q = run_layer(self.to_q, x)
            k, v = group(src)
        else:
            q = run_layer(self.to_q, x)
            k = run_layer(self.to_k, src)
            v = run_layer(self.to_v, src)
        o = F.attention(q, k, v, self.heads, qlay, qlay if klay is None else klay, self.scale)
        return run_layer(self.to_out[0], o, residual=residual)


def _effective_weight(mod):
    """(tag, make) of the weight a Linear leaf — or a cloneofsimo LoRA wrapper around one, dropout inactive, no selector — applies
    in a forward: W, or W + scale * up @ down (utils/lora.py:57-62 with the Dropout in eval mode).  None: not expressible."""
    if isinstance(mod, nn.Linear) and not hasattr(mod, "lora_A"):
        w = mod.weight
        # (a weight that trains moves under this library's own optimiser kernels without a version bump: functional.weights_epoch)
        return ("lin", w.data_ptr(), w._version, F.weights_epoch[0] if w.requires_grad else -1), (lambda: F.prepared_weight(w, "fwd")), mod.bias
    base = getattr(mod, "linear", None)
    if base is None or not (hasattr(mod, "lora_down") and hasattr(mod, "lora_up")) or _drop_p(getattr(mod, "dropout", None)) > 0.0:
        return None
    sel = getattr(mod, "selector", None)
    if sel is not None and not isinstance(sel, nn.Identity):
        return None
    w, u, d, sc = base.weight, mod.lora_up.weight, mod.lora_down.weight, float(mod.scale)
    if w.shape[0] % 8 or w.shape[1] % 8:
        return None
    tag = ("lora", w.data_ptr(), w._version, u.data_ptr(), u._version, d.data_ptr(), d._version, sc,
           F.weights_epoch[0] if (w.requires_grad or u.requires_grad or d.requires_grad) else -1)
    return tag, (lambda: F.prepared_weight(_folded_weight(mod, base), "fwd")), base.bias


def _temporal_unit_fused(norm, attn, t, qlay):
    """`t + attn(norm(t))` of a temporal BasicTransformerBlock as one launch (functional.temporal_attention_fused), or None when the
    one-launch form does not apply: rows not in the (batch, frame, pixel) order of TransformerTemporalModel, a width / clip length
    the library has no kernel for, projection biases, or a LoRA flavour whose forward is not `W + s up down`."""
    hw, frames = qlay.bdiv, qlay.S
    if not (qlay.ss == hw and qlay.lo == 1 and qlay.hi == frames * hw and qlay.nbatch % hw == 0):
        return None
    width = attn.heads * 64
    if t.shape[1] != width or not F.temporal_fused_ok(width, frames):
        return None
    parts = [_effective_weight(m) for m in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0])]
    if any(p is None for p in parts) or any(p[2] is not None for p in parts[:3]):
        return None
    if any(tuple(m.weight.shape if isinstance(m, nn.Linear) else m.linear.weight.shape) != (width, width)
           for m in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0])):
        return None
    tag = tuple(p[0] for p in parts)
    hit = attn.__dict__.get("_t2v_fused_w")
    if hit is None or hit[0] != tag or hit[1].device != t.device:
        hit = (tag, torch.cat([p[1]() for p in parts[:3]], dim=0).contiguous(), F.temporal_fused_prepare_wo(parts[3][1]()))
        attn.__dict__["_t2v_fused_w"] = hit
    return F.temporal_attention_fused(t, norm.weight, norm.bias, norm.eps, hit[1], hit[2], parts[3][2], qlay.nbatch // hw, frames, hw,
                                      attn.scale)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return F.geglu(run_layer(self.proj, x))


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim)])

    def forward(self, x, residual=None):
        return run_layer(self.net[2], self.net[0](x), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None, attention_bias=False,
                 double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim, bias=attention_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim, num_attention_heads,
                               attention_head_dim, bias=attention_bias)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, t, qlay, ctx=None, klay=None):
        if ctx is None and not torch.is_grad_enabled():
            # forward-only temporal block (sampling, validation): `norm -> attn -> + residual` as ONE launch each
            # (csrc/temporal_fused.hip) where the library has the kernel; the feed-forward keeps its launches
            for norm, attn in ((self.norm1, self.attn1), (self.norm2, self.attn2)):
                out = _temporal_unit_fused(norm, attn, t, qlay)
                if out is None:
                    n, r = F.layer_norm_res(t, norm.weight, norm.bias, norm.eps)
                    out = attn(n, qlay, residual=r)
                t = out
            n, r = F.layer_norm_res(t, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return self.ff(n, residual=r)
        # (normalised, pass-through) pairs: the residual's gradient is summed inside each LayerNorm's backward kernel
        n, r = F.layer_norm_res(t, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        t = self.attn1(n, qlay, residual=r)
        n, r = F.layer_norm_res(t, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        t = self.attn2(n, qlay, ctx, klay, residual=r)
        n, r = F.layer_norm_res(t, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return self.ff(n, residual=r)


class TextCtx:
    """encoder_hidden_states as a token matrix [B*S, D] (NOT repeated per frame: the F-fold
    `repeat_interleave` of models/unet_3d_condition.py:401 is expressed through the attention batch map)."""

    def __init__(self, ehs):
        b, s, d = ehs.shape
        self.m = ehs.reshape(b * s, d).to(BF16).contiguous()
        self.B, self.S = b, s


class Transformer2DModel(nn.Module):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 cross_attention_dim=None, norm_num_groups=32, use_linear_projection=True, **_):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states=None, num_frames=1, **_):
        hw = x.h * x.w
        a, xr = F.group_norm_res(x.m, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, False, x.n)
        t = run_layer(self.proj_in, a)
        qlay = SeqLayout(x.n, hw, hw, 0, 1, 1)
        ctx = encoder_hidden_states
        klay = SeqLayout(x.n, ctx.S, ctx.S, 0, 1, x.n // ctx.B)
        for blk in self.transformer_blocks:
            t = blk(t, qlay, ctx.m, klay)
        out = run_layer(self.proj_out, t, residual=xr, colsum=True)
        return _Out(sample=Tok(out, x.n, x.h, x.w))


class TransformerTemporalModel(nn.Module):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 cross_attention_dim=None, norm_num_groups=32, **_):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim,
                                  double_self_attention=True)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, encoder_hidden_states=None, num_frames=1, **_):
        hw = x.h * x.w
        B = x.n // num_frames
        a, xr = F.group_norm_res(x.m, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, False, B)
        t = run_layer(self.proj_in, a)
        # rows are (b, f, pixel): a sequence = the F rows of one pixel, stride hw rows — no permute copies
        qlay = SeqLayout(B * hw, num_frames, num_frames * hw, 1, hw, hw)
        for blk in self.transformer_blocks:
            t = blk(t, qlay)
        out = run_layer(self.proj_out, t, residual=xr, colsum=True)
        return _Out(sample=Tok(out, x.n, x.h, x.w))


# --------------------------------------------------------------------------- resampling
class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:   # VAE flavour: F.pad(x,(0,1,0,1)) then stride 2, pad 0
            cfg = ConvCfg("conv", x.n, x.h, x.w, 3, 3, 2, 0, 0, 0, x.h // 2, x.w // 2)
        else:
            cfg = ConvCfg.conv2d(x.n, x.h, x.w, 3, 2, self.padding)
        return Tok(run_layer(self.conv, x.m, cfg, colsum=True), x.n, cfg.Ho, cfg.Wo)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=True, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is not None and tuple(output_size) != (2 * x.h, 2 * x.w):
            raise RuntimeError("t2v_amd: only the exact 2x nearest upsample is implemented natively")
        cfg = ConvCfg.conv2d(x.n, x.h, x.w, 3, 1, 1, up=1)
        return Tok(run_layer(self.conv, x.m, cfg, colsum=True), x.n, 2 * x.h, 2 * x.w)
