"""Golden fixture for the diffusers -> ModelScope/webui key remap.

Runs the REFERENCE's `convert_unet_state_dict` (utils/convert_diffusers_to_original_ms_text_to_video.py:172-216, importable
here: it needs only torch + safetensors) on the drop-in UNet's state-dict keys and on a stable_lora LoRA state dict
(strict mapping), and records {hf_key: [ms_key, converted_shape]}.  Needs /root/reference; the fixture travels instead.
    python tests/golden/make_golden_keymap.py
"""
import contextlib
import io
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import t2v_amd  # noqa: E402,F401
from t2v_amd.models.unet_3d_condition import UNet3DConditionModel  # noqa: E402
from t2v_amd.stable_lora import lora as SL  # noqa: E402

sys.path.insert(0, "/root/reference")
with contextlib.redirect_stdout(io.StringIO()):
    from utils.convert_diffusers_to_original_ms_text_to_video import convert_unet_state_dict  # noqa: E402

SMALL = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=64)


def record(sd, strict):
    with contextlib.redirect_stdout(io.StringIO()):
        out = convert_unet_state_dict(dict(sd), strict_mapping=strict)
    # the converter keeps insertion order: mapping key i <-> output key i
    assert len(out) == len(sd)
    return {hf: [ms, list(out[ms].shape)] for hf, ms in zip(sd.keys(), out.keys())}


torch.manual_seed(0)
unet = UNet3DConditionModel(**SMALL)
full = record(unet.state_dict(), False)
SL.add_lora_to(unet, target_module=["Transformer2DModel", "ResnetBlock2D", "TransformerTemporalModel", "TemporalConvLayer"],
               search_class=[torch.nn.Linear, torch.nn.Conv2d, torch.nn.Conv3d], r=4)()
lora = record(SL.lora_state_dict(unet), True)
with open(os.path.join(os.path.dirname(__file__), "ms_keymap.json"), "w") as f:
    json.dump({"config": SMALL, "full": full, "stable_lora": lora}, f, indent=0)
print(len(full), "model keys,", len(lora), "LoRA keys")
