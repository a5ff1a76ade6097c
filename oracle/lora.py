"""cloneofsimo-style LoRA layers + injection, restated for the ORACLE (test infrastructure).

Follows the reference's vendored implementation:
  * layer forward  `y = base(x) + dropout(up(selector(down(x)))) * scale`  — utils/lora.py:57-62,134-139,211-216
  * init            down ~ N(0, (1/r)^2), up = 0; r clipped to min(in,out) — utils/lora.py:39-55,93-132,175-209
  * injection       class-name ancestor search, exact-class gate, shared base Parameters,
                    `parent._modules[name] = wrapper`                      — utils/lora.py:269-313,393-480
Pinned against the real reference by `tests/golden/make_golden.py` (fixtures) and, in the build
container, by direct import of `/root/reference/utils/lora.py` in `tests/test_oracle_lora.py`.
"""
import torch
from torch import nn


class _LoraBase(nn.Module):
    def forward(self, x):
        return self.base()(x) + self.dropout(self.lora_up(self.selector(self.lora_down(x)))) * self.scale


class LoraInjectedLinear(_LoraBase):
    def __init__(self, in_features, out_features, bias=False, r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        r = min(r, in_features, out_features)
        self.r = r
        self.linear = nn.Linear(in_features, out_features, bias)
        self.lora_down = nn.Linear(in_features, r, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Linear(r, out_features, bias=False)
        self.scale = scale
        self.selector = nn.Identity()
        nn.init.normal_(self.lora_down.weight, std=1 / r)
        nn.init.zeros_(self.lora_up.weight)

    def base(self):
        return self.linear


class LoraInjectedConv2d(_LoraBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        r = min(r, in_channels, out_channels)
        self.r = r
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.lora_down = nn.Conv2d(in_channels, r, kernel_size, stride, padding, dilation, groups, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv2d(r, out_channels, 1, 1, 0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        nn.init.normal_(self.lora_down.weight, std=1 / r)
        nn.init.zeros_(self.lora_up.weight)

    def base(self):
        return self.conv


class LoraInjectedConv3d(_LoraBase):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 1, 1), padding=(1, 0, 0), bias=False, r=4,
                 dropout_p=0.0, scale=1.0):
        super().__init__()
        r = min(r, in_channels, out_channels)
        self.r = r
        self.kernel_size, self.padding = kernel_size, padding
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, padding=padding)
        self.lora_down = nn.Conv3d(in_channels, r, kernel_size=kernel_size, bias=False, padding=padding)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv3d(r, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        nn.init.normal_(self.lora_down.weight, std=1 / r)
        nn.init.zeros_(self.lora_up.weight)

    def base(self):
        return self.conv


_WRAPPERS = (LoraInjectedLinear, LoraInjectedConv2d, LoraInjectedConv3d)


def find_modules(model, ancestor_class=None, search_class=(nn.Linear, nn.Conv2d, nn.Conv3d)):
    """utils/lora.py:269-313 (`_find_modules_v2`): yield (parent, name, module)."""
    if ancestor_class is not None:
        ancestors = [m for m in model.modules() if m.__class__.__name__ in ancestor_class]
    else:
        ancestors = list(model.modules())
    for anc in ancestors:
        for fullname, module in anc.named_modules():
            if isinstance(module, tuple(search_class)):
                *path, name = fullname.split(".")
                parent = anc
                for p in path:
                    parent = parent.get_submodule(p)
                if isinstance(parent, _WRAPPERS):
                    continue
                yield parent, name, module


def inject_trainable_lora_extended(model, target_replace_module=("UNet3DConditionModel",), r=4, loras=None):
    """utils/lora.py:393-480.  Returns (list of parameter generators, names)."""
    params, names = [], []
    if loras is not None:
        loras = torch.load(loras)
    for parent, name, child in find_modules(model, set(target_replace_module)):
        if child.__class__ == nn.Linear:
            w = LoraInjectedLinear(child.in_features, child.out_features, child.bias is not None, r=r)
            w.linear.weight = child.weight
            if child.bias is not None:
                w.linear.bias = child.bias
        elif child.__class__ == nn.Conv2d:
            w = LoraInjectedConv2d(child.in_channels, child.out_channels, child.kernel_size, child.stride,
                                   child.padding, child.dilation, child.groups, child.bias is not None, r=r)
            w.conv.weight = child.weight
            if child.bias is not None:
                w.conv.bias = child.bias
        elif child.__class__ == nn.Conv3d:
            w = LoraInjectedConv3d(child.in_channels, child.out_channels, bias=child.bias is not None,
                                   kernel_size=child.kernel_size, padding=child.padding, r=r)
            w.conv.weight = child.weight
            if child.bias is not None:
                w.conv.bias = child.bias
        else:
            continue
        w.to(child.weight.device).to(child.weight.dtype)
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
