"""LoRA bank: storage layout of the trainable LoRA factors inside the trainer's flat buffers.

The reference trains `lora_down` / `lora_up` of every wrapped layer (utils/lora.py:46-62,97-139,179-216).  Here every
factor is stored in the flat fp32 parameter buffer DIRECTLY in the layout the GEMM kernels consume
  down  [rp, taps, Cin_p]   (K ordered (tap, c); the Parameter is a permuted, sliced VIEW with the module's shape)
  up    [rp, Np]            (transposed: the backward `dt = dy U` reads it as a plain [N, K] weight and rides in the
                             base layer's backward-data launch; the forward `y += s t U^T` reads it K-major)
(rp / Np / Cin_p = sizes rounded up to 8, zero padded), so that
  * one cast kernel per step refreshes the bf16 copies of ALL factors (no per-layer permute/cast launches),
  * weight gradients are accumulated by the TN GEMM straight into the flat fp32 gradient buffer (no per-layer
    zeros / un-permute / AccumulateGrad launches), where clip + AdamW + the RCCL all-reduce operate.
"""
import torch

from .functional import ceil8


class LoraEntry:
    __slots__ = ("r", "rp", "n", "npad", "cin", "cin_p", "taps", "down_off", "up_off", "down_numel", "up_numel",
                 "down_w16", "up_w16", "down_g", "up_g", "flat_ptr")


def _wrapper_parts(mod):
    base = getattr(mod, "linear", None)
    if base is None:
        base = getattr(mod, "conv", None)
    if base is None or not hasattr(mod, "lora_down") or not hasattr(mod, "lora_up"):
        return None
    return base, mod.lora_down, mod.lora_up


def plan(model):
    """Map id(param) -> (entry, 'down'|'up') for every cloneofsimo-style wrapper in `model`."""
    plans = {}
    if model is None:
        return plans
    for mod in model.modules():
        parts = _wrapper_parts(mod)
        if parts is None:
            continue
        base, down, up = parts
        wd, wu = down.weight, up.weight
        if not (wd.requires_grad and wu.requires_grad):
            continue
        e = LoraEntry()
        e.r = wd.shape[0]
        e.rp = ceil8(e.r)
        e.n, e.npad = wu.shape[0], ceil8(wu.shape[0])
        e.cin, e.cin_p = wd.shape[1], ceil8(wd.shape[1])
        e.taps = 1
        for s in wd.shape[2:]:
            e.taps *= s
        if wd.dim() == 5 and (wd.shape[3] != 1 or wd.shape[4] != 1):
            continue    # only the (k,1,1) temporal conv is a supported 3-D window
        e.down_numel = e.rp * e.taps * e.cin_p
        e.up_numel = e.npad * e.rp
        plans[id(wd)] = (e, "down", mod)
        plans[id(wu)] = (e, "up", mod)
    return plans


def param_view(flat_slice, p, entry, role):
    """The tensor (shape == p.shape) through which the Parameter sees its slice of the flat buffer."""
    if role == "down":
        s3 = flat_slice.view(entry.rp, entry.taps, entry.cin_p)[: entry.r, :, : entry.cin]
        if p.dim() == 2:
            return s3[:, 0, :]
        if p.dim() == 4:
            return s3.reshape(entry.r, p.shape[2], p.shape[3], entry.cin).permute(0, 3, 1, 2) \
                if entry.cin == entry.cin_p else \
                flat_slice.view(entry.rp, p.shape[2], p.shape[3], entry.cin_p)[: entry.r, :, :, : entry.cin].permute(0, 3, 1, 2)
        v = s3.permute(0, 2, 1)
        return v[:, :, :, None, None]
    s2 = flat_slice.view(entry.rp, entry.npad)[: entry.r, : entry.n].t()
    for _ in range(p.dim() - 2):
        s2 = s2.unsqueeze(-1)
    return s2


def attach(plans, flat_p16, flat_g, offsets):
    """Create the prepared-layout views and hang the entry on its wrapper module (`mod._t2v_bank`)."""
    done = set()
    for pid, (e, role, mod) in plans.items():
        off = offsets.get(pid)
        if off is None:
            continue
        if role == "down":
            e.down_off = off
            e.down_w16 = flat_p16[off: off + e.down_numel].view(e.rp, e.taps * e.cin_p)
            e.down_g = flat_g[off: off + e.down_numel].view(e.rp, e.taps * e.cin_p)
        else:
            e.up_off = off
            e.up_w16 = flat_p16[off: off + e.up_numel].view(e.rp, e.npad)
            e.up_g = flat_g[off: off + e.up_numel].view(e.rp, e.npad)
        done.add(id(mod))
    for pid, (e, role, mod) in plans.items():
        if all(hasattr(e, a) for a in ("down_w16", "up_w16")):
            e.flat_ptr = flat_g.data_ptr()
            mod._t2v_bank = e
