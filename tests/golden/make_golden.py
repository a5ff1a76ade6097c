"""Generate golden fixtures from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports the executable parts of /root/reference: `utils/lora.py`
(LoRA layers + injection), `utils/bucketing.py`, and the state-dict key map of
`utils/convert_diffusers_to_original_ms_text_to_video.py`; writes small fixtures next to
this file.  The reference cannot travel to the GPU box; these fixtures can.
"""
import contextlib
import io
import json
import os
import sys

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from utils import lora as ref_lora  # noqa: E402  (reference code, executed as-is)
from utils.bucketing import sensible_buckets  # noqa: E402

from oracle.unet3d import UNet3DConditionModel  # noqa: E402

SMALL = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=64)


def lora_layer_fixtures():
    out = {}
    g = torch.Generator().manual_seed(42)
    specs = {
        "linear": (lambda: ref_lora.LoraInjectedLinear(64, 48, bias=True, r=4), (5, 7, 64)),
        "linear_nobias": (lambda: ref_lora.LoraInjectedLinear(128, 64, bias=False, r=16), (3, 128)),
        "conv2d": (lambda: ref_lora.LoraInjectedConv2d(16, 24, 3, 1, 1, r=4), (2, 16, 9, 11)),
        "conv2d_s2": (lambda: ref_lora.LoraInjectedConv2d(16, 16, 3, 2, 1, r=8), (2, 16, 8, 8)),
        "conv2d_1x1": (lambda: ref_lora.LoraInjectedConv2d(16, 32, 1, 1, 0, r=4), (2, 16, 5, 5)),
        "conv3d": (lambda: ref_lora.LoraInjectedConv3d(16, 16, (3, 1, 1), (1, 0, 0), r=4), (2, 16, 6, 5, 4)),
    }
    for name, (ctor, shape) in specs.items():
        torch.manual_seed(hash(name) % 1000)
        m = ctor().eval()
        with torch.no_grad():
            m.lora_up.weight.copy_(torch.randn(m.lora_up.weight.shape, generator=g) * 0.3)
        x = torch.randn(shape, generator=g)
        with torch.no_grad():
            y = m(x)
        out[name] = dict(state={k: v.clone() for k, v in m.state_dict().items()}, x=x, y=y, r=m.r, scale=m.scale)
    return out


def injection_fixture():
    torch.manual_seed(0)
    m = UNet3DConditionModel(**SMALL)
    with contextlib.redirect_stdout(io.StringIO()):
        params, names = ref_lora.inject_trainable_lora_extended(m, {"UNet3DConditionModel"}, r=4)
    wrapped = [n for n, mod in m.named_modules()
               if isinstance(mod, (ref_lora.LoraInjectedLinear, ref_lora.LoraInjectedConv2d,
                                   ref_lora.LoraInjectedConv3d))]
    kinds = {n: type(mod).__name__ for n, mod in m.named_modules() if n in set(wrapped)}
    n_lora = sum(p.numel() for gen in params for p in gen)
    # restricted target list as shipped in configs/v2/lora_training_config.yaml style
    torch.manual_seed(0)
    m2 = UNet3DConditionModel(**SMALL)
    with contextlib.redirect_stdout(io.StringIO()):
        _, names2 = ref_lora.inject_trainable_lora_extended(m2, {"Transformer2DModel", "ResnetBlock2D"}, r=4)
    return dict(count=len(names), wrapped=wrapped, kinds=kinds, lora_params=n_lora, count_t2d_resnet=len(names2))


def full_model_facts():
    with torch.device("meta"):
        m = UNet3DConditionModel()
    n_params = sum(p.numel() for p in m.parameters())
    layers = sum(1 for mod in m.modules() if type(mod) in (nn.Linear, nn.Conv2d, nn.Conv3d))
    sd = {k: torch.empty(v.shape, device="meta") for k, v in m.state_dict().items()}
    from utils.convert_diffusers_to_original_ms_text_to_video import convert_unet_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        conv = convert_unet_state_dict(sd)
    unmapped = sorted(k for k in conv if k.startswith(("down_blocks", "up_blocks", "mid_block", "conv_in",
                                                       "conv_out", "conv_norm_out", "time_embedding",
                                                       "transformer_in")))
    # LoRA sizes per rank (cloneofsimo): sum over layers of r'*(in*k + out), r' = min(r, in, out)
    def lora_size(r):
        tot = 0
        for mod in m.modules():
            if type(mod) is nn.Linear:
                rr = min(r, mod.in_features, mod.out_features)
                tot += rr * (mod.in_features + mod.out_features)
            elif type(mod) in (nn.Conv2d, nn.Conv3d):
                rr = min(r, mod.in_channels, mod.out_channels)
                k = 1
                for s in mod.kernel_size:
                    k *= s
                tot += rr * (mod.in_channels * k + mod.out_channels)
        return tot
    return dict(n_params=n_params, lora_layers=layers, n_keys=len(sd), converted_keys=sorted(conv.keys()),
                unmapped_hf_keys=unmapped, diffusers_keys=sorted(sd.keys()),
                lora_size={str(r): lora_size(r) for r in (4, 16, 32)})


def bucket_fixture():
    cases = [(384, 384, 1280, 720), (512, 512, 1280, 720), (576, 320, 1920, 1080), (1024, 576, 1920, 1080),
             (256, 256, 720, 1280), (256, 256, 512, 512), (512, 320, 1080, 1920), (384, 256, 640, 480)]
    return [dict(args=list(c), out=list(sensible_buckets(*c))) for c in cases]


if __name__ == "__main__":
    torch.save(lora_layer_fixtures(), os.path.join(HERE, "lora_layers.pt"))
    with open(os.path.join(HERE, "lora_injection.json"), "w") as f:
        json.dump(injection_fixture(), f, indent=0)
    with open(os.path.join(HERE, "unet_facts.json"), "w") as f:
        json.dump(full_model_facts(), f, indent=0)
    with open(os.path.join(HERE, "buckets.json"), "w") as f:
        json.dump(bucket_fixture(), f, indent=0)
    print("golden fixtures written to", HERE)
