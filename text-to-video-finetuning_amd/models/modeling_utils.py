"""Minimal stand-in for the diffusers ModelMixin/ConfigMixin surface that `train.py` and
`utils/lora_handler.py` use on the UNet: `.config`, `.dtype`, `.device`,
`from_pretrained(path, subfolder=)`, `save_pretrained(dir)`, `from_config`,
`enable/disable_gradient_checkpointing`, `enable_xformers_memory_efficient_attention` (no-op: the native
attention core is always used), and survival of `copy.deepcopy`, `.cpu()`, `.to(device, dtype)`."""
import json
import os
from types import SimpleNamespace

import torch


class FrozenConfig(SimpleNamespace):
    def to_dict(self):
        return {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self).items()}

    def __getitem__(self, k):
        return getattr(self, k)


class ModelMixinLite:
    config_name = "config.json"
    _supports_gradient_checkpointing = True

    def register_to_config(self, **kw):
        self.config = FrozenConfig(**kw)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_gradient_checkpointing(self):
        self._set_gradient_checkpointing(value=True)

    def disable_gradient_checkpointing(self):
        self._set_gradient_checkpointing(value=False)

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        return None

    def set_attn_processor(self, processor):
        for m in self.modules():
            if hasattr(m, "set_processor"):
                m.set_processor(processor)

    def _convert_checkpoint_keys(self, sd):
        """Hook: map a stock diffusers checkpoint onto this model's keys (identity by default; the load stays strict)."""
        return sd

    @classmethod
    def from_config(cls, config):
        cfg = config.to_dict() if hasattr(config, "to_dict") else dict(config)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        return cls(**cfg)

    def save_pretrained(self, save_directory, safe_serialization=True, **_):
        os.makedirs(save_directory, exist_ok=True)
        cfg = self.config.to_dict()
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "diffusion_pytorch_model.bin"))

    @classmethod
    def _ctor_kwargs_from_checkpoint(cls, sd):
        """Hook: constructor arguments that follow from what the checkpoint holds (none by default)."""
        return {}

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, torch_dtype=None, **overrides):
        d = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        st = os.path.join(d, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "diffusion_pytorch_model.bin"), map_location="cpu")
        import inspect
        allowed = set(inspect.signature(cls.__init__).parameters) - {"self"}
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in allowed}
        kw.update(cls._ctor_kwargs_from_checkpoint(sd))
        kw.update({k: v for k, v in overrides.items() if k in allowed})
        model = cls(**kw)
        sd = model._convert_checkpoint_keys(sd)   # per-class key filter / remap of stock diffusers checkpoints
        model.load_state_dict(sd, strict=True)
        if torch_dtype is not None:
            model.to(torch_dtype)
        return model
