"""`stable_lora` flavour of LoRA for the drop-in UNet — mirror of the reference's `stable_lora/lora.py`.

The reference builds on `loralib` (un-vendored, absent offline); the few loralib pieces it uses are restated here
(SURVEY.md Appendix A.9): `LoRALayer`, `Linear` (`y = x W^T + drop(x) A^T B^T * alpha/r`, A kaiming-uniform, B zeros),
`mark_only_lora_as_trainable`, `lora_state_dict`.  The reference's own layers are mirrored with the same semantics:
  Conv2d / Conv3d re-materialise `W + (B @ A).view(...) * scaling` every forward     stable_lora/lora.py:69-197
  (Conv3d: `.view(out, in, k, k, 1)` then `mean(dim=-2)`, merge force-disabled       :148-149,176-197)
  find_modules / add_lora_to (shares weight + bias, `module._modules[name] = l`)      :27-67,257-302
  save_lora (full-weights + webui safetensors) / load_lora / set_mode_group            :304-387
On the device path the parents evaluate these layers through `models.leaves.run_layer` (attributes `lora_A`,
`lora_B`, `scaling`, `merged`): the effective weight is formed once per call and the implicit-GEMM kernels run on it.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

UNET_REPLACE = ["Transformer2DModel", "ResnetBlock2D"]
TEXT_ENCODER_REPLACE = ["CLIPAttention", "CLIPTextEmbeddings"]
UNET_ATTENTION_REPLACE = ["CrossAttention"]
TEXT_ENCODER_ATTENTION_REPLACE = ["CLIPAttention", "CLIPTextEmbeddings"]


class LoRALayer:
    def __init__(self, r, lora_alpha, lora_dropout, merge_weights):
        self.r = r
        self.lora_alpha = lora_alpha
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else (lambda x: x)
        self.merged = False
        self.merge_weights = merge_weights


def _init_factors(layer):
    if hasattr(layer, "lora_A"):
        nn.init.kaiming_uniform_(layer.lora_A, a=math.sqrt(5))
        nn.init.zeros_(layer.lora_B)


class Linear(nn.Linear, LoRALayer):
    def __init__(self, in_features, out_features, r=0, lora_alpha=1, lora_dropout=0.0, merge_weights=True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        LoRALayer.__init__(self, r, lora_alpha, lora_dropout, merge_weights)
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        _init_factors(self)

    def delta(self):
        return (self.lora_B @ self.lora_A) * self.scaling

    def train(self, mode=True):
        nn.Linear.train(self, mode)
        if self.merge_weights and self.r > 0 and self.merged == mode:
            self.weight.data += self.delta() * (-1 if mode else 1)     # (`.data +=` does not bump the version counter:
            self.weight.__dict__.pop("_t2v_prep", None)                # drop the cached bf16 GEMM copies of the old values)
            self.merged = not mode
        return self

    def forward(self, x):
        y = F.linear(x, self.weight, self.bias)
        if self.r > 0 and not self.merged:
            y = y + (self.lora_dropout(x) @ self.lora_A.t() @ self.lora_B.t()) * self.scaling
        return y


class Conv2d(nn.Conv2d, LoRALayer):
    def __init__(self, in_channels, out_channels, kernel_size, r=0, lora_alpha=1, lora_dropout=0.0, merge_weights=True,
                 **kwargs):
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, **kwargs)
        LoRALayer.__init__(self, r, lora_alpha, lora_dropout, merge_weights)
        assert type(kernel_size) is int
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * kernel_size, in_channels * kernel_size)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_channels * kernel_size, r * kernel_size)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        _init_factors(self)

    def delta(self):
        return (self.lora_B @ self.lora_A).view(self.weight.shape) * self.scaling

    def train(self, mode=True):
        nn.Conv2d.train(self, mode)
        if self.merge_weights and self.r > 0 and self.merged == mode:
            self.weight.data += self.delta() * (-1 if mode else 1)     # (`.data +=` does not bump the version counter:
            self.weight.__dict__.pop("_t2v_prep", None)                # drop the cached bf16 GEMM copies of the old values)
            self.merged = not mode
        return self

    def forward(self, x):
        if self.r > 0 and not self.merged:
            return F.conv2d(x, self.weight + self.delta(), self.bias, self.stride, self.padding, self.dilation, self.groups)
        return nn.Conv2d.forward(self, x)


class Conv3d(nn.Conv3d, LoRALayer):
    def __init__(self, in_channels, out_channels, kernel_size, r=0, lora_alpha=1, lora_dropout=0.0, merge_weights=True,
                 **kwargs):
        nn.Conv3d.__init__(self, in_channels, out_channels, (kernel_size, 1, 1), **kwargs)
        LoRALayer.__init__(self, r, lora_alpha, lora_dropout, merge_weights)
        assert type(kernel_size) is int
        i, o, k = self.weight.shape[:3]
        self.view_shape = (i, o, k, kernel_size, 1)
        self.force_disable_merge = True
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * kernel_size, in_channels * kernel_size)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_channels * kernel_size, r * kernel_size)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        _init_factors(self)

    def delta(self):
        return torch.mean((self.lora_B @ self.lora_A).view(self.view_shape), dim=-2, keepdim=True) * self.scaling

    def forward(self, x):
        if self.r > 0 and not self.merged:
            return F.conv3d(x, self.weight + self.delta(), self.bias, self.stride, self.padding, self.dilation, self.groups)
        return nn.Conv3d.forward(self, x)


class Embedding(nn.Embedding, LoRALayer):
    """loralib's Embedding (what `create_lora_emb`, stable_lora/lora.py:241-248, builds for CLIPTextEmbeddings' token and
    position tables): `E[x] + (A^T[x] @ B^T) * alpha/r`, A [r, num_embeddings] zeros, B [dim, r] normal.  The text encoder runs
    through stock torch ops, so this layer is evaluated by its own forward."""

    def __init__(self, num_embeddings, embedding_dim, r=0, lora_alpha=1, merge_weights=True, **kwargs):
        nn.Embedding.__init__(self, num_embeddings, embedding_dim, **kwargs)
        LoRALayer.__init__(self, r, lora_alpha, 0.0, merge_weights)
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r, num_embeddings)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((embedding_dim, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
            nn.init.zeros_(self.lora_A)
            nn.init.normal_(self.lora_B)

    def delta(self):
        return (self.lora_B @ self.lora_A).t() * self.scaling

    def train(self, mode=True):
        nn.Embedding.train(self, mode)
        if self.merge_weights and self.r > 0 and self.merged == mode:
            self.weight.data += self.delta() * (-1 if mode else 1)
            self.merged = not mode
        return self

    def forward(self, x):
        y = nn.Embedding.forward(self, x)
        if self.r > 0 and not self.merged:
            a = F.embedding(x, self.lora_A.t(), self.padding_idx, self.max_norm, self.norm_type, self.scale_grad_by_freq, self.sparse)
            y = y + (a @ self.lora_B.t()) * self.scaling
        return y


_LORA_TYPES = (Linear, Conv2d, Conv3d, Embedding)


def mark_only_lora_as_trainable(model, bias="none"):
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False
    if bias == "all":
        for n, p in model.named_parameters():
            if "bias" in n:
                p.requires_grad = True
    elif bias == "lora_only":
        for m in model.modules():
            if isinstance(m, LoRALayer) and getattr(m, "bias", None) is not None:
                m.bias.requires_grad = True


def lora_state_dict(model, bias="none"):
    sd = model.state_dict()
    out = {k: v for k, v in sd.items() if "lora_" in k}
    if bias == "all":
        out.update({k: v for k, v in sd.items() if "bias" in k})
    return out


def find_modules(model, ancestor_class=None, search_class=(nn.Linear,), exclude_children_of=_LORA_TYPES):
    if ancestor_class is not None:
        ancestors = [m for m in model.modules() if m.__class__.__name__ in ancestor_class]
    else:
        ancestors = list(model.modules())
    for anc in ancestors:
        for fullname, module in anc.named_modules():
            if not isinstance(module, tuple(search_class)):
                continue
            *path, name = fullname.split(".")
            parent = anc
            for p in path:
                parent = parent.get_submodule(p)
            if exclude_children_of and isinstance(parent, tuple(exclude_children_of)):
                continue
            yield parent, name, module


def activate_lora_train(model, bias):
    def unfreeze():
        print(model.__class__.__name__ + " LoRA set for training.")
        return mark_only_lora_as_trainable(model, bias=bias)
    return unfreeze


def add_lora_to(model, target_module=UNET_REPLACE, search_class=(nn.Linear,), r=32, dropout=0, lora_bias="none"):
    for parent, name, child in list(find_modules(model, ancestor_class=target_module, search_class=search_class)):
        if isinstance(child, _LORA_TYPES):
            continue
        has_bias = getattr(child, "bias", None) is not None
        common = dict(merge_weights=False, bias=has_bias, lora_dropout=dropout, lora_alpha=r, r=r)
        if isinstance(child, nn.Linear):
            l = Linear(child.in_features, child.out_features, **common)
        elif isinstance(child, nn.Conv2d):
            l = Conv2d(child.in_channels, child.out_channels, kernel_size=child.kernel_size[0], padding=child.padding,
                       stride=child.stride, **common)
        elif isinstance(child, nn.Conv3d):
            l = Conv3d(child.in_channels, child.out_channels, kernel_size=child.kernel_size[0], padding=child.padding,
                       stride=child.stride, **common)
        elif isinstance(child, nn.Embedding):            # create_lora_emb (stable_lora/lora.py:241-248, 287-288)
            l = Embedding(child.num_embeddings, child.embedding_dim, merge_weights=False, lora_alpha=r, r=r)
        else:
            continue
        if has_bias:
            l.bias = child.bias
        l.weight = child.weight
        l.to(child.weight.device)
        parent._modules[name] = l
    return activate_lora_train(model, lora_bias)


def save_lora(unet=None, text_encoder=None, save_text_weights=False, output_dir="output", lora_filename="lora.safetensors",
              lora_bias="none", save_for_webui=True, only_webui=False, metadata=None, unet_dict_converter=None,
              text_dict_converter=None):
    """stable_lora/lora.py:304-362: `full_weights/{name}_unet.safetensors` (+ `_text_encoder`) in fp32 for further finetuning,
    and `webui_{name}.safetensors`: the UNet LoRA keys renamed to the original ModelScope layout (converter), fp16."""
    from safetensors.torch import save_file
    ext = ".safetensors"
    if not only_webui:
        d = os.path.join(output_dir, "full_weights")
        os.makedirs(d, exist_ok=True)
        base = os.path.join(d, lora_filename)
        for i, model in enumerate([unet, text_encoder]):
            if model is None or (i == 1 and not save_text_weights):
                continue
            sd = {k: v.detach().cpu().contiguous() for k, v in lora_state_dict(model, lora_bias).items()}
            save_file(sd, base + ("_text_encoder" if i == 1 else "_unet") + ext)
    if save_for_webui and unet is not None:
        if unet_dict_converter is None:
            from ..utils.convert_diffusers_to_original_ms_text_to_video import convert_unet_state_dict as unet_dict_converter
        out = unet_dict_converter(lora_state_dict(unet, lora_bias), strict_mapping=True)
        if save_text_weights and text_encoder is not None:
            if text_dict_converter is None:
                from ..utils.convert_diffusers_to_original_ms_text_to_video import convert_text_enc_state_dict as text_dict_converter
            out.update(text_dict_converter(lora_state_dict(text_encoder, lora_bias)))
        out = {k: v.detach().to(torch.float16).cpu().contiguous() for k, v in out.items()}
        os.makedirs(output_dir, exist_ok=True)
        save_file(out, os.path.join(output_dir, f"webui_{lora_filename}{ext}"), metadata=metadata)


def load_lora(model, lora_path):
    try:
        if os.path.exists(lora_path):
            from safetensors.torch import load_file
            model.load_state_dict(load_file(lora_path), strict=False)
    except Exception as e:   # noqa: BLE001
        print(f"Could not load your lora file: {e}")


def set_mode(model, train=False):
    for m in model.modules():
        if hasattr(m, "merged"):
            m.train(train)


def set_mode_group(models, train):
    for model in models:
        set_mode(model, train)
        model.train(train)
