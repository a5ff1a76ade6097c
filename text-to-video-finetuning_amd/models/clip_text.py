"""The CLIP text tower of the train step (`encoder_hidden_states = text_encoder(token_ids)[0]`, train.py:784-790) on the native
kernels: SURVEY 8(f) row 2.

The module tree stays the `transformers.CLIPTextModel` the reference loads (`train.py:120-121`: state-dict keys, `.to()`,
`requires_grad_`, and the reference's text-LoRA injection into `CLIPEncoderLayer`'s Linear layers, `utils/lora_handler.py`, all keep
working on it); only its FORWARD is re-expressed here: per layer LayerNorm -> q/k/v Linear -> causal self-attention (16 heads x 64
for the ModelScope / OpenCLIP ViT-H text config) -> out Linear (+residual) -> LayerNorm -> fc1 -> GELU -> fc2 (+residual),
then the final LayerNorm — every op a launch of this library (`functional.conv_linear` through `leaves.run_layer`, so LoRA wrappers
swapped in for the Linear layers are evaluated like everywhere else; `layer_norm`; `attention(causal=True)`; `gelu`).  The two
embedding lookups stay torch index ops (a gather is not a kernel of this path).
"""
import os

import torch

from .. import functional as F
from ..functional import SeqLayout
from .leaves import run_layer

BF16 = torch.bfloat16
_native = os.environ.get("T2V_NATIVE_CLIP", "1") != "0"       # A/B switch: 0 = call the transformers module as it is


def _text_model(model):
    return getattr(model, "text_model", model)        # (transformers < 5 nests the tower one level down)


def supported(model):
    """True if `model` is a CLIP text tower this forward covers: head_dim 64, gelu / quick_gelu, the HF attribute names."""
    tm = _text_model(model)
    cfg = getattr(model, "config", None)
    if cfg is None or not all(hasattr(tm, a) for a in ("embeddings", "encoder", "final_layer_norm")):
        return False
    heads, hidden = getattr(cfg, "num_attention_heads", 0), getattr(cfg, "hidden_size", 0)
    if heads <= 0 or hidden != heads * 64 or getattr(cfg, "hidden_act", None) not in ("gelu", "quick_gelu"):
        return False
    layers = getattr(tm.encoder, "layers", None)
    return layers is not None and all(hasattr(l, a) for l in layers for a in ("layer_norm1", "self_attn", "layer_norm2", "mlp"))


def _qkv_fused(a):
    """(weight [3C, C], bias [3C]) of a frozen, unwrapped attention's q/k/v projections as ONE layer (cached on the module, validated
    against the three Parameters' storage and version): one launch instead of three on a 77-row problem that is all launch latency."""
    mods = (a.q_proj, a.k_proj, a.v_proj)
    if not all(type(m) is torch.nn.Linear and m.bias is not None and not m.weight.requires_grad and not m.bias.requires_grad for m in mods):
        return None
    tag = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods)
    hit = a.__dict__.get("_t2v_qkv")
    if hit is not None and hit[0] != tag and hit[1].device == mods[0].weight.device and hit[1].dtype == mods[0].weight.dtype \
            and hit[1].shape[0] == sum(m.weight.shape[0] for m in mods) and hit[1].shape[1] == mods[0].weight.shape[1]:
        # a member changed (load_state_dict / resume): re-concatenate INTO THE SAME BUFFERS — a captured step keeps reading them,
        # and the in-place copy bumps their version so the derived bf16 / fp32 copies follow (functional.resync_prepared)
        with torch.no_grad():
            hit[1].copy_(torch.cat([m.weight.detach() for m in mods], 0))
            hit[2].copy_(torch.cat([m.bias.detach() for m in mods], 0))
        hit = (tag, hit[1], hit[2])
        a.__dict__["_t2v_qkv"] = hit
    elif hit is None or hit[0] != tag:
        w = torch.nn.Parameter(torch.cat([m.weight.detach() for m in mods], 0).contiguous(), requires_grad=False)
        b = torch.nn.Parameter(torch.cat([m.bias.detach() for m in mods], 0).contiguous(), requires_grad=False)
        hit = (tag, w, b)
        a.__dict__["_t2v_qkv"] = hit
    return hit[1], hit[2]


def fused_params(model, refresh=True):
    """The concatenated q/k/v Parameters `_qkv_fused` keeps in the attention modules' `__dict__` (they are not in
    `model.parameters()`): the trainer lists them with the frozen parameters whose cached GEMM copies a captured step reads, and
    calls this before every replay so that re-loaded q/k/v weights reach the fused copy."""
    out = []
    if model is None or not supported(model):
        return out
    for l in _text_model(model).encoder.layers:
        a = l.self_attn
        if "_t2v_qkv" in a.__dict__:
            hit = _qkv_fused(a) if refresh else a.__dict__["_t2v_qkv"][1:]
            if hit is not None:
                out += [hit[0], hit[1]]
    return out


def text_states(model, input_ids):
    """`model(input_ids)[0]` — final-LayerNorm'ed hidden states [B, S, hidden], bf16 — through the native kernels."""
    tm = _text_model(model)
    cfg = model.config
    ids = input_ids.view(-1, input_ids.shape[-1])
    if not ids.is_cuda:
        raise RuntimeError("t2v_amd: the native CLIP forward runs on a ROCm device only")
    B, S = ids.shape
    x = tm.embeddings(input_ids=ids)                  # token + position embeddings (two gathers and an add)
    hidden, heads = x.shape[-1], cfg.num_attention_heads
    x = x.reshape(B * S, hidden).to(BF16).contiguous()
    lay = SeqLayout(B, S, S, 0, 1)
    kind = 0 if cfg.hidden_act == "gelu" else 1
    eps = float(cfg.layer_norm_eps)
    for l in tm.encoder.layers:
        a = l.self_attn
        h = F.layer_norm(x, l.layer_norm1.weight, l.layer_norm1.bias, eps)
        fused = _qkv_fused(a) if not torch.is_grad_enabled() or not h.requires_grad else None
        if fused is not None:
            qkv = F.conv_linear(h, fused[0], fused[1])
            q, k, v = qkv[:, :hidden], qkv[:, hidden:2 * hidden], qkv[:, 2 * hidden:]
        else:
            q, k, v = run_layer(a.q_proj, h), run_layer(a.k_proj, h), run_layer(a.v_proj, h)
        o = F.attention(q, k, v, heads, lay, lay, float(getattr(a, "scale", 0.125)), causal=True)
        x = run_layer(a.out_proj, o, residual=x)
        h = F.layer_norm(x, l.layer_norm2.weight, l.layer_norm2.bias, eps)
        h = F.gelu(run_layer(l.mlp.fc1, h), kind)
        x = run_layer(l.mlp.fc2, h, residual=x)
    x = F.layer_norm(x, tm.final_layer_norm.weight, tm.final_layer_norm.bias, eps)
    return x.view(B, S, hidden)


_warned = set()


def encode(model, input_ids):
    """What the trainer calls: the native forward where it applies, else the module itself (`model(ids)[0]`) — with ONE warning per
    reason: that path runs the vendor GEMM / attention kernels of stock PyTorch-ROCm, not this library's."""
    if _native and input_ids.is_cuda and supported(model):
        return text_states(model, input_ids)
    why = ("T2V_NATIVE_CLIP=0" if not _native else "token ids on the host" if not input_ids.is_cuda
           else f"{type(model).__name__}: not a CLIP text tower with head_dim 64 and gelu / quick_gelu")
    if why not in _warned:
        _warned.add(why)
        import warnings
        warnings.warn(f"t2v_amd: the text encoder runs through its own (stock PyTorch-ROCm) forward, not the native kernels: {why}",
                      RuntimeWarning, stacklevel=2)
    return model(input_ids)[0]
