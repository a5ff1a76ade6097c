"""diffusers -> original ModelScope / webui key names for the UNet3D.

Same result as the reference's `convert_unet_state_dict` (utils/convert_diffusers_to_original_ms_text_to_video.py:172-216) —
pinned key by key against it through `tests/golden/ms_keymap.json` — written as a direct function of the parsed key instead of
an ordered chain of substring replacements:

  time_embedding.linear_{1,2}          -> time_embed.{0,2}
  conv_in / conv_norm_out / conv_out   -> input_blocks.0.0 / out.0 / out.2
  transformer_in                       -> input_blocks.0.1
  down_blocks.i.<kind>.j               -> input_blocks.(3i+j+1).<slot>        up_blocks.i.<kind>.j -> output_blocks.(3i+j).<slot>
  mid_block.<kind>.j                   -> middle_block.<slot'>
      kind: resnets -> slot 0, attentions -> 1, temp_attentions -> 2, temp_convs -> "0.temopral_conv" (sic, the original's typo);
      mid: resnets.j -> 3j, attentions -> 1, temp_attentions -> 2, temp_convs.j -> "3j.temopral_conv"
  down_blocks.i.downsamplers.0.conv    -> input_blocks.3(i+1).op               up_blocks.i.upsamplers.0 -> output_blocks.(3i+2).(1 if i==0 else 3)
  inside a ResnetBlock2D: norm1/conv1/norm2/conv2/time_emb_proj/conv_shortcut -> in_layers.0/in_layers.2/out_layers.0/out_layers.3/
                                                                                  emb_layers.1/skip_connection
Tensor quirk kept: every `proj_in`/`proj_out` non-bias tensor gains a trailing unit dimension (the temporal transformers of the
original are Conv1d) EXCEPT the exact `.weight` tensors of the spatial transformers (slot 1) — so LoRA factors of a spatial
`proj_in` do get it (that is what the reference produces for a LoRA state dict).
"""
import re

_TOP = {
    "time_embedding.linear_1": "time_embed.0", "time_embedding.linear_2": "time_embed.2",
    "conv_in": "input_blocks.0.0", "conv_norm_out": "out.0", "conv_out": "out.2",
}
_RESNET = (("norm1", "in_layers.0"), ("conv1", "in_layers.2"), ("norm2", "out_layers.0"), ("conv2", "out_layers.3"),
           ("time_emb_proj", "emb_layers.1"), ("conv_shortcut", "skip_connection"))
_BLOCK = re.compile(r"^(down_blocks|up_blocks)\.(\d+)\.(resnets|attentions|temp_convs|temp_attentions)\.(\d+)\.(.*)$")
_MID = re.compile(r"^mid_block\.(resnets|attentions|temp_convs|temp_attentions)\.(\d+)\.(.*)$")
_DOWNS = re.compile(r"^down_blocks\.(\d+)\.downsamplers\.0\.conv\.(.*)$")
_UPS = re.compile(r"^up_blocks\.(\d+)\.upsamplers\.0\.(.*)$")
_SLOT = {"resnets": "0", "attentions": "1", "temp_attentions": "2", "temp_convs": "0.temopral_conv"}


def _resnet_tail(tail):
    for hf, ms in _RESNET:
        tail = tail.replace(hf, ms)
    return tail


def ms_key(key):
    """ModelScope / webui name of one diffusers UNet3D state-dict key (unknown keys are returned unchanged)."""
    for hf, ms in _TOP.items():
        if key.startswith(hf + "."):
            return ms + key[len(hf):]
    if key.startswith("transformer_in."):
        return "input_blocks.0.1." + key[len("transformer_in."):]
    m = _BLOCK.match(key)
    if m:
        side, i, kind, j, tail = m.group(1), int(m.group(2)), m.group(3), int(m.group(4)), m.group(5)
        if kind == "resnets":
            tail = _resnet_tail(tail)
        idx = 3 * i + j + 1 if side == "down_blocks" else 3 * i + j
        return f"{'input_blocks' if side == 'down_blocks' else 'output_blocks'}.{idx}.{_SLOT[kind]}.{tail}"
    m = _MID.match(key)
    if m:
        kind, j, tail = m.group(1), int(m.group(2)), m.group(3)
        if kind == "resnets":
            return f"middle_block.{3 * j}.{_resnet_tail(tail)}"
        if kind == "temp_convs":
            return f"middle_block.{3 * j}.temopral_conv.{tail}"
        return f"middle_block.{1 if kind == 'attentions' else 2}.{tail}"
    m = _DOWNS.match(key)
    if m:
        return f"input_blocks.{3 * (int(m.group(1)) + 1)}.op.{m.group(2)}"
    m = _UPS.match(key)
    if m:
        i = int(m.group(1))
        return f"output_blocks.{3 * i + 2}.{1 if i == 0 else 3}.{m.group(2)}"
    return key


# spatial transformers sit in slot 1 of input_blocks 1.. / output_blocks 3.. / middle_block (input_blocks.0.1 is transformer_in,
# a TEMPORAL transformer, and keeps the Conv1d-style trailing dimension)
_SPATIAL_PROJ = re.compile(r"^(input_blocks\.[1-9]\d*|output_blocks\.([3-9]|1\d)|middle_block)\.1\.proj_(in|out)\.weight$")


def convert_unet_state_dict(unet_state_dict, strict_mapping=False):
    """{ms_key: tensor}; `strict_mapping` is accepted for signature parity (keys absent from the dict are never invented
    here, which is what the reference's strict mode guarantees)."""
    out = {}
    for k, v in unet_state_dict.items():
        nk = ms_key(k)
        if "proj_" in k and "bias" not in k and not _SPATIAL_PROJ.match(nk):
            v = v.unsqueeze(-1)
        out[nk] = v
    return out


def convert_text_enc_state_dict(text_enc_dict):
    """The CLIP text encoder keeps its keys (reference: identity, :298-299)."""
    return text_enc_dict
