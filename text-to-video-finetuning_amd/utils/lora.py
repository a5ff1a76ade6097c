"""cloneofsimo-style LoRA for the drop-in UNet — the host-side mirror of the reference's `utils/lora.py`.

Same public names and semantics (cited per symbol) so `utils/lora_handler.py`-style callers work unchanged:
  LoraInjectedLinear / LoraInjectedConv2d / LoraInjectedConv3d      utils/lora.py:33-237
  _find_modules (= _find_modules_v2)                                utils/lora.py:269-313
  inject_trainable_lora / inject_trainable_lora_extended            utils/lora.py:336-480
  extract_lora_ups_down, save_lora_weight                           utils/lora.py:530-582
  monkeypatch_or_replace_lora_extended (loader)                     utils/lora.py:862-982
  collapse_lora, monkeypatch_remove_lora                            utils/lora.py:781-815,998-1047
The wrapper classes are ordinary torch modules (their forward composes `base(x) + dropout(up(selector(down(x)))) * scale`),
so CPU use works as in the reference; on the device path the native parent modules never call `wrapper(x)` —
`models.leaves.run_layer` recognises the wrapper and evaluates the same expression with the HIP GEMM kernels.
The reference's own `utils/lora.py` can equally be used to inject into the drop-in model (tested).
"""
from itertools import groupby
from typing import List, Optional, Set, Type

import torch
from torch import nn

UNET_DEFAULT_TARGET_REPLACE = {"CrossAttention", "Attention", "GEGLU"}
UNET_EXTENDED_TARGET_REPLACE = {"ResnetBlock2D", "CrossAttention", "Attention", "GEGLU"}
TEXT_ENCODER_DEFAULT_TARGET_REPLACE = {"CLIPAttention"}
TEXT_ENCODER_EXTENDED_TARGET_REPLACE = {"CLIPAttention"}
DEFAULT_TARGET_REPLACE = UNET_DEFAULT_TARGET_REPLACE


def _clip_rank(r, a, b):
    lim = min(a, b)
    if r > lim:
        print(f"LoRA rank {r} is too large. setting to: {lim}")
        return lim
    return r


class _LoraMixin:
    def _base(self):
        return self.linear if hasattr(self, "linear") else self.conv

    def forward(self, input):
        return self._base()(input) + self.dropout(self.lora_up(self.selector(self.lora_down(input)))) * self.scale

    def realize_as_lora(self):
        return self.lora_up.weight.data * self.scale, self.lora_down.weight.data

    def _init_factors(self):
        nn.init.normal_(self.lora_down.weight, std=1 / self.r)
        nn.init.zeros_(self.lora_up.weight)

    def _diag_selector(self, make, diag):
        assert diag.shape == (self.r,)
        self.selector = make()
        w = torch.diag(diag).to(self.lora_up.weight.device).to(self.lora_up.weight.dtype)
        self.selector.weight.data = w.view(self.selector.weight.shape) if self.selector.weight.dim() > 2 else w


class LoraInjectedLinear(_LoraMixin, nn.Module):
    def __init__(self, in_features, out_features, bias=False, r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        self.r = _clip_rank(r, in_features, out_features)
        self.linear = nn.Linear(in_features, out_features, bias)
        self.lora_down = nn.Linear(in_features, self.r, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Linear(self.r, out_features, bias=False)
        self.scale = scale
        self.selector = nn.Identity()
        self._init_factors()

    def set_selector_from_diag(self, diag):
        self._diag_selector(lambda: nn.Linear(self.r, self.r, bias=False), diag)


class LoraInjectedConv2d(_LoraMixin, nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        self.r = _clip_rank(r, in_channels, out_channels)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.lora_down = nn.Conv2d(in_channels, self.r, kernel_size, stride, padding, dilation, groups, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv2d(self.r, out_channels, 1, 1, 0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        self._init_factors()

    def set_selector_from_diag(self, diag):
        self._diag_selector(lambda: nn.Conv2d(self.r, self.r, 1, 1, 0, bias=False), diag)


class LoraInjectedConv3d(_LoraMixin, nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 1, 1), padding=(1, 0, 0), bias=False, r=4,
                 dropout_p=0, scale=1.0):
        super().__init__()
        self.r = _clip_rank(r, in_channels, out_channels)
        self.kernel_size, self.padding = kernel_size, padding
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, padding=padding)
        self.lora_down = nn.Conv3d(in_channels, self.r, kernel_size=kernel_size, bias=False, padding=padding)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv3d(self.r, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        self._init_factors()

    def set_selector_from_diag(self, diag):
        self._diag_selector(lambda: nn.Conv3d(self.r, self.r, 1, 1, 0, bias=False), diag)


_WRAPPERS = (LoraInjectedLinear, LoraInjectedConv2d, LoraInjectedConv3d)


def _find_modules_v2(model, ancestor_class: Optional[Set[str]] = None,
                     search_class: List[Type[nn.Module]] = (nn.Linear,),
                     exclude_children_of: Optional[List[Type[nn.Module]]] = _WRAPPERS):
    """Yield (parent, name, module) for every `search_class` descendant of modules whose class NAME is in
    `ancestor_class`, skipping children of LoRA wrappers."""
    if ancestor_class is not None:
        ancestors = [m for m in model.modules() if m.__class__.__name__ in ancestor_class]
    else:
        ancestors = list(model.modules())
    search = tuple(search_class)
    for anc in ancestors:
        for fullname, module in anc.named_modules():
            if not isinstance(module, search):
                continue
            *path, name = fullname.split(".")
            parent = anc
            for p in path:
                parent = parent.get_submodule(p)
            if exclude_children_of and isinstance(parent, tuple(exclude_children_of)):
                continue
            yield parent, name, module


_find_modules = _find_modules_v2


def _wrap(child, r, dropout_p=None, scale=1.0):
    """Build the wrapper for an exact nn.Linear / nn.Conv2d / nn.Conv3d (subclasses are skipped, as in the reference)."""
    kw = {} if dropout_p is None else {"dropout_p": dropout_p}
    if child.__class__ == nn.Linear:
        w = LoraInjectedLinear(child.in_features, child.out_features, child.bias is not None, r=r, scale=scale, **kw)
        w.linear.weight = child.weight
        if child.bias is not None:
            w.linear.bias = child.bias
    elif child.__class__ == nn.Conv2d:
        w = LoraInjectedConv2d(child.in_channels, child.out_channels, child.kernel_size, child.stride, child.padding,
                               child.dilation, child.groups, child.bias is not None, r=r, scale=scale, **kw)
        w.conv.weight = child.weight
        if child.bias is not None:
            w.conv.bias = child.bias
    elif child.__class__ == nn.Conv3d:
        w = LoraInjectedConv3d(child.in_channels, child.out_channels, bias=child.bias is not None,
                               kernel_size=child.kernel_size, padding=child.padding, r=r, scale=scale, **kw)
        w.conv.weight = child.weight
        if child.bias is not None:
            w.conv.bias = child.bias
    else:
        return None
    return w.to(child.weight.device).to(child.weight.dtype)


def _inject(model, targets, search, r, loras, dropout_p=None, scale=1.0):
    params, names = [], []
    if loras is not None:
        loras = torch.load(loras)
    for parent, name, child in _find_modules(model, targets, search_class=search):
        w = _wrap(child, r, dropout_p, scale)
        if w is None:
            continue
        parent._modules[name] = w
        params.append(w.lora_up.parameters())
        params.append(w.lora_down.parameters())
        if loras is not None:
            w.lora_up.weight = loras.pop(0)
            w.lora_down.weight = loras.pop(0)
        w.lora_up.weight.requires_grad = True
        w.lora_down.weight.requires_grad = True
        names.append(name)
    return params, names


def inject_trainable_lora(model, target_replace_module=DEFAULT_TARGET_REPLACE, r=4, loras=None, verbose=False,
                          dropout_p=0.0, scale=1.0):
    return _inject(model, set(target_replace_module), [nn.Linear], r, loras, dropout_p, scale)


def inject_trainable_lora_extended(model, target_replace_module=UNET_EXTENDED_TARGET_REPLACE, r=4, loras=None):
    return _inject(model, set(target_replace_module), [nn.Linear, nn.Conv2d, nn.Conv3d], r, loras)


def extract_lora_ups_down(model, target_replace_module=DEFAULT_TARGET_REPLACE):
    loras = [(m.lora_up, m.lora_down) for _, _, m in
             _find_modules(model, set(target_replace_module), search_class=list(_WRAPPERS), exclude_children_of=None)]
    if not loras:
        raise ValueError("No lora injected.")
    return loras


def save_lora_weight(model, path="./lora.pt", target_replace_module=DEFAULT_TARGET_REPLACE):
    """`.pt` = flat list [up0, down0, up1, down1, ...] (utils/lora.py:570-582)."""
    weights = []
    for up, down in extract_lora_ups_down(model, target_replace_module):
        weights.append(up.weight.to("cpu").to(torch.float32))
        weights.append(down.weight.to("cpu").to(torch.float32))
    torch.save(weights, path)


def monkeypatch_or_replace_lora_extended(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE, r=4):
    """Load factors from the flat list `loras` into (newly injected or existing) wrappers, in module order."""
    if isinstance(r, int):
        r_of = lambda: r
    else:
        it = iter(r)
        r_of = lambda: next(it)
    for parent, name, child in _find_modules(model, set(target_replace_module),
                                             search_class=[nn.Linear, nn.Conv2d, nn.Conv3d, *_WRAPPERS]):
        if isinstance(child, _WRAPPERS):
            w = child
            r_of()
        else:
            w = _wrap(child, r_of())
            if w is None:
                continue
            parent._modules[name] = w
        up, down = loras.pop(0), loras.pop(0)
        w.lora_up.weight = nn.Parameter(up.type(w._base().weight.dtype).to(w._base().weight.device))
        w.lora_down.weight = nn.Parameter(down.type(w._base().weight.dtype).to(w._base().weight.device))


def collapse_lora(model, replace_modules=UNET_EXTENDED_TARGET_REPLACE | TEXT_ENCODER_EXTENDED_TARGET_REPLACE, alpha=1.0):
    """Fold `alpha * up @ down` into the base weight in place (utils/lora.py:781-815)."""
    for _, _, m in _find_modules(model, set(replace_modules), search_class=list(_WRAPPERS), exclude_children_of=None):
        base = m._base()
        up = m.lora_up.weight.data.flatten(1)
        down = m.lora_down.weight.data.flatten(1)
        delta = (up @ down).reshape(base.weight.shape)
        base.weight = nn.Parameter(base.weight.data + alpha * delta.type(base.weight.dtype).to(base.weight.device))


def monkeypatch_remove_lora(model):
    """Replace every wrapper by its base layer (utils/lora.py:998-1047)."""
    for parent, name, m in list(_find_modules(model, None, search_class=list(_WRAPPERS), exclude_children_of=None)):
        parent._modules[name] = m._base()


def tune_lora_scale(model, alpha=1.0):
    for m in model.modules():
        if isinstance(m, _WRAPPERS):
            m.scale = alpha


def set_lora_diag(model, diag):
    for m in model.modules():
        if isinstance(m, _WRAPPERS):
            m.set_selector_from_diag(diag)
