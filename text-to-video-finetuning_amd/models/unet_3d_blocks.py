"""Native counterparts of the five block classes of the reference's `models/unet_3d_blocks.py:267-875`.

Same class names, constructor arguments, attribute names (`resnets`, `temp_convs`, `attentions`,
`temp_attentions`, `downsamplers`, `upsamplers`, `gradient_checkpointing`, `has_cross_attention`) and the same
order of sub-ops (down/up: resnet -> temp_conv -> attn -> temp_attn; mid: attn -> temp_attn -> resnet ->
temp_conv, with the un-guarded first temp_conv of `:387`).  Activations flow as channels-last token matrices
(`leaves.Tok`); `torch.cat` of skip tensors (`:764,861`) is a strided two-source copy kernel.
"""
from torch import nn

from .. import functional as F
from .leaves import (Downsample2D, ResnetBlock2D, TemporalConvLayer, Tok, Transformer2DModel, TransformerTemporalModel,
                     Upsample2D)


def _resnet(cin, cout, temb, eps, groups, scale=1.0):
    return ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups,
                         output_scale_factor=scale)


def _cat(h, skip):
    return Tok(F.concat(h.m, skip.m), h.n, h.h, h.w)


def checkpointed(fn, *args, **kwargs):
    """`torch.utils.checkpoint(..., use_reentrant=False)` around one sub-module call — the reference's per-sub-module
    recompute (`custom_checkpoint` / `cross_attn_g_c` / `up_down_g_c` / `transformer_g_c`, models/unet_3d_blocks.py:30-153):
    nothing the call saves for backward is kept; the call is re-run when its gradient is needed.  The dropout seed counter
    (models/leaves.py) is rewound for the re-run so that it regenerates the masks of the first run."""
    from torch.utils.checkpoint import checkpoint
    from . import leaves
    start = leaves._seed_state["ctr"]

    def run(*a, **k):
        later = leaves._seed_state["ctr"]
        leaves._seed_state["ctr"] = start
        out = fn(*a, **k)
        leaves._seed_state["ctr"] = max(later, leaves._seed_state["ctr"])
        return out

    return checkpoint(run, *args, use_reentrant=False, **kwargs)


def _call(block, mod, *args, **kwargs):
    """Run sub-module `mod` of `block`, recomputed in backward when the block's `gradient_checkpointing` flag is set
    (train.py:127-129,670-675 -> `_set_gradient_checkpointing`)."""
    import torch
    if getattr(block, "gradient_checkpointing", False) and torch.is_grad_enabled():
        return checkpointed(mod, *args, **kwargs)
    return mod(*args, **kwargs)


class UNetMidBlock3DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 attn_num_head_channels=1, output_scale_factor=1.0, cross_attention_dim=1280,
                 dual_cross_attention=False, use_linear_projection=True, upcast_attention=False):
        super().__init__()
        self.gradient_checkpointing = False
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        c, hd = in_channels, attn_num_head_channels
        resnets = [_resnet(c, c, temb_channels, resnet_eps, resnet_groups, output_scale_factor)]
        temp_convs = [TemporalConvLayer(c, c, dropout=0.1)]
        attentions, temp_attentions = [], []
        for _ in range(num_layers):
            attentions.append(Transformer2DModel(c // hd, hd, in_channels=c, num_layers=1,
                                                 cross_attention_dim=cross_attention_dim,
                                                 norm_num_groups=resnet_groups))
            temp_attentions.append(TransformerTemporalModel(c // hd, hd, in_channels=c, num_layers=1,
                                                            cross_attention_dim=cross_attention_dim,
                                                            norm_num_groups=resnet_groups))
            resnets.append(_resnet(c, c, temb_channels, resnet_eps, resnet_groups, output_scale_factor))
            temp_convs.append(TemporalConvLayer(c, c, dropout=0.1))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.attentions = nn.ModuleList(attentions)
        self.temp_attentions = nn.ModuleList(temp_attentions)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None, num_frames=1,
                cross_attention_kwargs=None):
        hidden_states = _call(self, self.resnets[0], hidden_states, temb)
        hidden_states = _call(self, self.temp_convs[0], hidden_states, num_frames=num_frames)
        for attn, temp_attn, resnet, temp_conv in zip(self.attentions, self.temp_attentions, self.resnets[1:],
                                                      self.temp_convs[1:]):
            hidden_states = _call(self, attn, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  num_frames=num_frames).sample
            if num_frames > 1:
                hidden_states = _call(self, temp_attn, hidden_states, num_frames=num_frames).sample
            hidden_states = _call(self, resnet, hidden_states, temb)
            if num_frames > 1:
                hidden_states = _call(self, temp_conv, hidden_states, num_frames=num_frames)
        return hidden_states


class CrossAttnDownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0, downsample_padding=1,
                 add_downsample=True, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False):
        super().__init__()
        self.gradient_checkpointing = False
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        hd = attn_num_head_channels
        resnets, temp_convs, attentions, temp_attentions = [], [], [], []
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            resnets.append(_resnet(cin, out_channels, temb_channels, resnet_eps, resnet_groups))
            temp_convs.append(TemporalConvLayer(out_channels, out_channels, dropout=0.1))
            attentions.append(Transformer2DModel(out_channels // hd, hd, in_channels=out_channels, num_layers=1,
                                                 cross_attention_dim=cross_attention_dim,
                                                 norm_num_groups=resnet_groups))
            temp_attentions.append(TransformerTemporalModel(out_channels // hd, hd, in_channels=out_channels,
                                                            num_layers=1, cross_attention_dim=cross_attention_dim,
                                                            norm_num_groups=resnet_groups))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.attentions = nn.ModuleList(attentions)
        self.temp_attentions = nn.ModuleList(temp_attentions)
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                            padding=downsample_padding, name="op")])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None, num_frames=1,
                cross_attention_kwargs=None):
        output_states = ()
        for resnet, temp_conv, attn, temp_attn in zip(self.resnets, self.temp_convs, self.attentions,
                                                      self.temp_attentions):
            hidden_states = _call(self, resnet, hidden_states, temb)
            if num_frames > 1:
                hidden_states = _call(self, temp_conv, hidden_states, num_frames=num_frames)
            hidden_states = _call(self, attn, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  num_frames=num_frames).sample
            if num_frames > 1:
                hidden_states = _call(self, temp_attn, hidden_states, num_frames=num_frames).sample
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class DownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, dropout=0.0, num_layers=1, resnet_eps=1e-6,
                 resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32, resnet_pre_norm=True,
                 output_scale_factor=1.0, add_downsample=True, downsample_padding=1):
        super().__init__()
        self.gradient_checkpointing = False
        resnets, temp_convs = [], []
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            resnets.append(_resnet(cin, out_channels, temb_channels, resnet_eps, resnet_groups))
            temp_convs.append(TemporalConvLayer(out_channels, out_channels, dropout=0.1))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                            padding=downsample_padding, name="op")])

    def forward(self, hidden_states, temb=None, num_frames=1):
        output_states = ()
        for resnet, temp_conv in zip(self.resnets, self.temp_convs):
            hidden_states = _call(self, resnet, hidden_states, temb)
            if num_frames > 1:
                hidden_states = _call(self, temp_conv, hidden_states, num_frames=num_frames)
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class CrossAttnUpBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, dropout=0.0, num_layers=1,
                 resnet_eps=1e-6, resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32,
                 resnet_pre_norm=True, attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0,
                 add_upsample=True, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False):
        super().__init__()
        self.gradient_checkpointing = False
        self.has_cross_attention = True
        self.attn_num_head_channels = attn_num_head_channels
        hd = attn_num_head_channels
        resnets, temp_convs, attentions, temp_attentions = [], [], [], []
        for i in range(num_layers):
            res_skip = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(_resnet(rin + res_skip, out_channels, temb_channels, resnet_eps, resnet_groups))
            temp_convs.append(TemporalConvLayer(out_channels, out_channels, dropout=0.1))
            attentions.append(Transformer2DModel(out_channels // hd, hd, in_channels=out_channels, num_layers=1,
                                                 cross_attention_dim=cross_attention_dim,
                                                 norm_num_groups=resnet_groups))
            temp_attentions.append(TransformerTemporalModel(out_channels // hd, hd, in_channels=out_channels,
                                                            num_layers=1, cross_attention_dim=cross_attention_dim,
                                                            norm_num_groups=resnet_groups))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.attentions = nn.ModuleList(attentions)
        self.temp_attentions = nn.ModuleList(temp_attentions)
        self.upsamplers = None
        if add_upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                upsample_size=None, attention_mask=None, num_frames=1, cross_attention_kwargs=None):
        for resnet, temp_conv, attn, temp_attn in zip(self.resnets, self.temp_convs, self.attentions,
                                                      self.temp_attentions):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = _call(self, resnet, _cat(hidden_states, res), temb)
            if num_frames > 1:
                hidden_states = _call(self, temp_conv, hidden_states, num_frames=num_frames)
            hidden_states = _call(self, attn, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  num_frames=num_frames).sample
            if num_frames > 1:
                hidden_states = _call(self, temp_attn, hidden_states, num_frames=num_frames).sample
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class UpBlock3D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, dropout=0.0, num_layers=1,
                 resnet_eps=1e-6, resnet_time_scale_shift="default", resnet_act_fn="swish", resnet_groups=32,
                 resnet_pre_norm=True, output_scale_factor=1.0, add_upsample=True):
        super().__init__()
        self.gradient_checkpointing = False
        resnets, temp_convs = [], []
        for i in range(num_layers):
            res_skip = in_channels if (i == num_layers - 1) else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(_resnet(rin + res_skip, out_channels, temb_channels, resnet_eps, resnet_groups))
            temp_convs.append(TemporalConvLayer(out_channels, out_channels, dropout=0.1))
        self.resnets = nn.ModuleList(resnets)
        self.temp_convs = nn.ModuleList(temp_convs)
        self.upsamplers = None
        if add_upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, num_frames=1):
        for resnet, temp_conv in zip(self.resnets, self.temp_convs):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = _call(self, resnet, _cat(hidden_states, res), temb)
            if num_frames > 1:
                hidden_states = _call(self, temp_conv, hidden_states, num_frames=num_frames)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                   resnet_act_fn, attn_num_head_channels, resnet_groups=None, cross_attention_dim=None,
                   downsample_padding=None, **_):
    if down_block_type == "DownBlock3D":
        return DownBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                           temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
                           resnet_groups=resnet_groups, downsample_padding=downsample_padding)
    if down_block_type == "CrossAttnDownBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock3D")
        return CrossAttnDownBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                                    temb_channels=temb_channels, add_downsample=add_downsample, resnet_eps=resnet_eps,
                                    resnet_groups=resnet_groups, downsample_padding=downsample_padding,
                                    cross_attention_dim=cross_attention_dim,
                                    attn_num_head_channels=attn_num_head_channels)
    raise ValueError(f"{down_block_type} does not exist.")


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, resnet_eps, resnet_act_fn, attn_num_head_channels, resnet_groups=None,
                 cross_attention_dim=None, **_):
    if up_block_type == "UpBlock3D":
        return UpBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                         prev_output_channel=prev_output_channel, temb_channels=temb_channels,
                         add_upsample=add_upsample, resnet_eps=resnet_eps, resnet_groups=resnet_groups)
    if up_block_type == "CrossAttnUpBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlock3D")
        return CrossAttnUpBlock3D(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
                                  prev_output_channel=prev_output_channel, temb_channels=temb_channels,
                                  add_upsample=add_upsample, resnet_eps=resnet_eps, resnet_groups=resnet_groups,
                                  cross_attention_dim=cross_attention_dim,
                                  attn_num_head_channels=attn_num_head_channels)
    raise ValueError(f"{up_block_type} does not exist.")
