"""`LoraHandler` — mirror of the reference facade `utils/lora_handler.py:70-351` for the cloneofsimo flavour.

Same constructor, `add_lora_to_model(use, model, replace_modules, dropout, lora_path, r) -> (params, negation)`
(`:239-268`), `save_lora_weights` (`:335-351`, `{step}_unet.pt` list-of-tensors format) and
`deactivate_lora_train` (`:271-277`).  The stable_lora flavour needs `loralib` (absent offline); wrappers in that
style that are already in a model are still evaluated natively (`models.leaves.run_layer`).
"""
import os
from types import SimpleNamespace

import torch

from .lora import inject_trainable_lora_extended, monkeypatch_or_replace_lora_extended, save_lora_weight

LoraVersions = SimpleNamespace(stable_lora="stable_lora", cloneofsimo="cloneofsimo")
LoraFuncTypes = SimpleNamespace(loader="loader", injector="injector")
FILE_BASENAMES = ["unet", "text_encoder"]
LORA_FILE_TYPES = [".pt", ".safetensors"]


class LoraHandler(object):
    def __init__(self, version=LoraVersions.cloneofsimo, use_unet_lora=False, use_text_lora=False, save_for_webui=False,
                 only_for_webui=False, lora_bias="none", unet_replace_modules=("UNet3DConditionModel",),
                 text_encoder_replace_modules=("CLIPEncoderLayer",)):
        if version not in (LoraVersions.cloneofsimo, LoraVersions.stable_lora):
            raise ValueError(f"unknown LoRA version {version}")
        self.version = version
        if version == LoraVersions.cloneofsimo:
            self.lora_loader = monkeypatch_or_replace_lora_extended
            self.lora_injector = inject_trainable_lora_extended
        else:
            from ..stable_lora.lora import add_lora_to, load_lora
            self.lora_loader, self.lora_injector = load_lora, add_lora_to
        self.lora_bias = lora_bias
        self.use_unet_lora, self.use_text_lora = use_unet_lora, use_text_lora
        self.save_for_webui, self.only_for_webui = save_for_webui, only_for_webui
        self.unet_replace_modules = list(unet_replace_modules)
        self.text_encoder_replace_modules = list(text_encoder_replace_modules)
        self.use_lora = any([use_text_lora, use_unet_lora])
        if self.use_lora:
            print(f"Using LoRA Version: {self.version}")

    def is_cloneofsimo_lora(self):
        return self.version == LoraVersions.cloneofsimo

    def is_stable_lora(self):
        return self.version == LoraVersions.stable_lora

    @staticmethod
    def _is_unet(model):
        return model.__class__.__name__ == "UNet3DConditionModel"

    def get_lora_file_path(self, lora_path, model):
        if lora_path is None or not os.path.exists(lora_path):
            return None
        want = FILE_BASENAMES[0] if self._is_unet(model) else FILE_BASENAMES[1]
        for f in sorted(os.listdir(lora_path)):
            base, ext = os.path.splitext(f)
            if ext in LORA_FILE_TYPES and want in base:
                return os.path.join(lora_path, f)
        return None

    def add_lora_to_model(self, use_lora, model, replace_modules, dropout=0.0, lora_path=None, r=16):
        """Returns (params, negation): params = list of parameter generators (cloneofsimo) or the model itself."""
        params, negation = None, None
        if use_lora and self.is_stable_lora():       # utils/lora_handler.py:188-237 (stable_lora branch)
            import torch
            lora_file = self.get_lora_file_path(lora_path, model)
            activator = self.lora_injector(model, target_module=list(replace_modules),
                                           search_class=[torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Embedding], r=r, dropout=dropout,
                                           lora_bias=self.lora_bias)
            activator()
            if lora_file is not None:
                self.lora_loader(model, lora_file)
            return model, None
        if use_lora:
            lora_file = self.get_lora_file_path(lora_path, model)
            params, negation = self.lora_injector(model, target_replace_module=set(replace_modules), r=r, loras=lora_file)
            n = sum(1 for _ in model.modules() if _.__class__.__name__.startswith("LoraInjected"))
            print(f"Successfully injected LoRA into {n} layers of {model.__class__.__name__}.")
        params = model if params is None else params
        return params, negation

    def deactivate_lora_train(self, models, deactivate=True):
        """utils/lora_handler.py:271-277: only the stable_lora flavour has a train/eval mode to toggle around sampling."""
        if self.is_stable_lora():
            from ..stable_lora.lora import set_mode_group
            set_mode_group(models, not deactivate)

    def save_lora_weights(self, model, save_path="", step=""):
        """utils/lora_handler.py:279-351: files go to `{save_path}/lora/`; cloneofsimo: `{step}_unet.pt` /
        `{step}_text_encoder.pt`, each only if that LoRA is in use; stable_lora: `full_weights/` + `webui_` safetensors."""
        save_path = os.path.join(save_path, "lora")
        os.makedirs(save_path, exist_ok=True)
        unet = getattr(model, "unet", model)
        te = getattr(model, "text_encoder", None)
        if self.is_stable_lora():
            import uuid
            from ..stable_lora.lora import save_lora
            name = "lora_text_to_video"
            save_lora(unet=unet, text_encoder=te, save_text_weights=self.use_text_lora, output_dir=save_path,
                      lora_filename=f"{step}_{name}", lora_bias=self.lora_bias, save_for_webui=self.save_for_webui,
                      only_webui=self.only_for_webui,
                      metadata={"stable_lora_text_to_video": "v1", "lora_name": name + "_" + uuid.uuid4().hex.lower()[:5]})
            return
        if any([self.save_for_webui, self.only_for_webui]):
            import warnings
            warnings.warn("'save_for_webui' is only supported by the stable_lora flavour (utils/lora_handler.py:340-346)")
        if self.use_unet_lora and self.unet_replace_modules is not None:
            save_lora_weight(unet, os.path.join(save_path, f"{step}_unet.pt"), set(self.unet_replace_modules))
        if te is not None and self.use_text_lora and self.text_encoder_replace_modules is not None:
            save_lora_weight(te, os.path.join(save_path, f"{step}_text_encoder.pt"), set(self.text_encoder_replace_modules))
