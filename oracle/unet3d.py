"""CPU fp32 restatement of the UNet3DConditionModel hot path (ORACLE — test infrastructure).

The reference wires diffusers leaf modules together; the leaf arithmetic is not in
the reference repository.  This file restates both:

* leaf operators (published diffusers ~v0.17 semantics):
  Timesteps, TimestepEmbedding, ResnetBlock2D, TemporalConvLayer, Attention,
  GEGLU/FeedForward, BasicTransformerBlock, Transformer2DModel,
  TransformerTemporalModel, Downsample2D, Upsample2D;
* the reference's own wiring: `models/unet_3d_blocks.py:267-875` (five block
  classes) and `models/unet_3d_condition.py:86-251,325-500` (model).

Class names, attribute names and state-dict keys equal the reference/diffusers
ones so that `utils/lora.py:_find_modules_v2` (class-name ancestor search) and the
key map in `utils/convert_diffusers_to_original_ms_text_to_video.py:18-169` apply.
Everything runs in plain torch ops; no product code is imported.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn


class _Out(SimpleNamespace):
    """Tiny stand-in for diffusers' *Output dataclasses (`.sample`)."""


# --------------------------------------------------------------------------- leaf ops
class Timesteps(nn.Module):
    # models/unet_3d_condition.py:138 -> Timesteps(320, True, 0); SURVEY Appendix A.1
    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    # models/unet_3d_condition.py:141-145
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class ResnetBlock2D(nn.Module):
    # built at models/unet_3d_blocks.py:295-306 etc.; SURVEY Appendix A.2
    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, eps=1e-5, groups=32,
                 dropout=0.0, time_embedding_norm="default", non_linearity="silu",
                 output_scale_factor=1.0, pre_norm=True):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, input_tensor, temb=None):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if temb is not None and self.time_emb_proj is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class TemporalConvLayer(nn.Module):
    # built at models/unet_3d_blocks.py:308-314 (dropout=0.1); SURVEY Appendix A.3
    def __init__(self, in_dim, out_dim=None, dropout=0.0):
        super().__init__()
        out_dim = out_dim or in_dim
        self.in_dim, self.out_dim = in_dim, out_dim
        self.conv1 = nn.Sequential(
            nn.GroupNorm(32, in_dim), nn.SiLU(), nn.Conv3d(in_dim, out_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv2 = nn.Sequential(
            nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
            nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv3 = nn.Sequential(
            nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
            nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv4 = nn.Sequential(
            nn.GroupNorm(32, out_dim), nn.SiLU(), nn.Dropout(dropout),
            nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        nn.init.zeros_(self.conv4[-1].weight)  # upstream: identity at init
        nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, hidden_states, num_frames=1):
        x = hidden_states[None, :].reshape((-1, num_frames) + hidden_states.shape[1:]).permute(0, 2, 1, 3, 4)
        identity = x
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        x = identity + x
        return x.permute(0, 2, 1, 3, 4).reshape((x.shape[0] * x.shape[2], -1) + x.shape[3:])


class Attention(nn.Module):
    # diffusers Attention + AttnProcessor2_0 (train.py:138-139); SURVEY Appendix A.6
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def set_processor(self, processor):  # train.py:139 — the reference's kernel-swap seam
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b, s, _ = hidden_states.shape
        q = self.to_q(hidden_states).view(b, s, self.heads, self.dim_head).transpose(1, 2)
        k = self.to_k(ctx).view(b, ctx.shape[1], self.heads, self.dim_head).transpose(1, 2)
        v = self.to_v(ctx).view(b, ctx.shape[1], self.heads, self.dim_head).transpose(1, 2)
        p = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(b, s, self.heads * self.dim_head)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states):
        h, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim)])

    def forward(self, hidden_states):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None,
                 attention_bias=False, double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, num_attention_heads, attention_head_dim, bias=attention_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, None if double_self_attention else cross_attention_dim,
                               num_attention_heads, attention_head_dim, bias=attention_bias)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states=None, **_):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states) + hidden_states
        return self.ff(self.norm3(hidden_states)) + hidden_states


class Transformer2DModel(nn.Module):
    # built at models/unet_3d_blocks.py:479-490 with use_linear_projection=True; Appendix A.4
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 cross_attention_dim=None, norm_num_groups=32, use_linear_projection=True, **_):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, *args, **kwargs):
        n, c, h, w = hidden_states.shape
        residual = hidden_states
        x = self.norm(hidden_states).permute(0, 2, 3, 1).reshape(n, h * w, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states)
        x = self.proj_out(x).reshape(n, h, w, c).permute(0, 3, 1, 2)
        return _Out(sample=x + residual)


class TransformerTemporalModel(nn.Module):
    # built at models/unet_3d_blocks.py:491-500, models/unet_3d_condition.py:147-152; Appendix A.5
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 cross_attention_dim=None, norm_num_groups=32, **_):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim,
                                  double_self_attention=True)
            for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, num_frames=1, **_):
        bf, c, h, w = hidden_states.shape
        b = bf // num_frames
        residual = hidden_states
        x = hidden_states[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.norm(x)
        x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, num_frames, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=None)
        x = self.proj_out(x)
        x = x[None, None, :].reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).reshape(bf, c, h, w)
        return _Out(sample=x + residual)


class Downsample2D(nn.Module):
    # models/unet_3d_blocks.py:506-513 -> Downsample2D(C, use_conv=True, out_channels=C, padding=1, name="op")
    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        if self.padding == 0:  # VAE encoder flavour
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    # models/unet_3d_blocks.py:742,852 -> Upsample2D(C, use_conv=True, out_channels=C)
    def __init__(self, channels, use_conv=True, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        return self.conv(hidden_states)


# --------------------------------------------------------------------------- blocks (reference wiring)
def _resnet(cin, cout, temb, eps, groups, scale=1.0):
    return ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=temb, eps=eps, groups=groups,
                         output_scale_factor=scale)


class UNetMidBlock3DCrossAttn(nn.Module):
    # models/unet_3d_blocks.py:267-419
    def __init__(self, in_channels, temb_channels, resnet_eps=1e-6, resnet_groups=32, attn_num_head_channels=1,
                 output_scale_factor=1.0, cross_attention_dim=1280, num_layers=1):
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

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                num_frames=1, cross_attention_kwargs=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.temp_convs[0](hidden_states, num_frames=num_frames)  # no F>1 guard (:387)
        for attn, temp_attn, resnet, temp_conv in zip(self.attentions, self.temp_attentions,
                                                      self.resnets[1:], self.temp_convs[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            if num_frames > 1:
                hidden_states = temp_attn(hidden_states, num_frames=num_frames).sample
            hidden_states = resnet(hidden_states, temb)
            if num_frames > 1:
                hidden_states = temp_conv(hidden_states, num_frames=num_frames)
        return hidden_states


class CrossAttnDownBlock3D(nn.Module):
    # models/unet_3d_blocks.py:422-569
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 attn_num_head_channels=1, cross_attention_dim=1280, downsample_padding=1, add_downsample=True):
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

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                num_frames=1, cross_attention_kwargs=None):
        output_states = ()
        for resnet, temp_conv, attn, temp_attn in zip(self.resnets, self.temp_convs, self.attentions,
                                                      self.temp_attentions):
            hidden_states = resnet(hidden_states, temb)
            if num_frames > 1:
                hidden_states = temp_conv(hidden_states, num_frames=num_frames)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            if num_frames > 1:
                hidden_states = temp_attn(hidden_states, num_frames=num_frames).sample
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class DownBlock3D(nn.Module):
    # models/unet_3d_blocks.py:572-652
    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6, resnet_groups=32,
                 add_downsample=True, downsample_padding=1):
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
            hidden_states = resnet(hidden_states, temb)
            if num_frames > 1:
                hidden_states = temp_conv(hidden_states, num_frames=num_frames)
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class CrossAttnUpBlock3D(nn.Module):
    # models/unet_3d_blocks.py:655-798
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, attn_num_head_channels=1, cross_attention_dim=1280, add_upsample=True):
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
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
            if num_frames > 1:
                hidden_states = temp_conv(hidden_states, num_frames=num_frames)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states).sample
            if num_frames > 1:
                hidden_states = temp_attn(hidden_states, num_frames=num_frames).sample
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class UpBlock3D(nn.Module):
    # models/unet_3d_blocks.py:801-875
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1, resnet_eps=1e-6,
                 resnet_groups=32, add_upsample=True):
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
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
            if num_frames > 1:
                hidden_states = temp_conv(hidden_states, num_frames=num_frames)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


# --------------------------------------------------------------------------- model
MODELSCOPE_CONFIG = dict(
    sample_size=32, in_channels=4, out_channels=4,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
    mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1024,
    attention_head_dim=64)


class UNet3DConditionModel(nn.Module):
    """models/unet_3d_condition.py:53-500 restated (ctor :86-251, forward :325-500)."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4,
                 down_block_types=MODELSCOPE_CONFIG["down_block_types"],
                 up_block_types=MODELSCOPE_CONFIG["up_block_types"],
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1024, attention_head_dim=64):
        super().__init__()
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            downsample_padding=downsample_padding, mid_block_scale_factor=mid_block_scale_factor, act_fn=act_fn,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
            attention_head_dim=attention_head_dim)
        self.sample_size = sample_size
        self.gradient_checkpointing = False
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[0], kernel_size=3, padding=1)
        time_embed_dim = boc[0] * 4
        self.time_proj = Timesteps(boc[0], True, 0)
        self.time_embedding = TimestepEmbedding(boc[0], time_embed_dim, act_fn=act_fn)
        self.transformer_in = TransformerTemporalModel(num_attention_heads=8, attention_head_dim=attention_head_dim,
                                                       in_channels=boc[0], num_layers=1)
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * len(down_block_types)
        out_ch = boc[0]
        for i, t in enumerate(down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            final = i == len(boc) - 1
            if t == "CrossAttnDownBlock3D":
                blk = CrossAttnDownBlock3D(in_ch, out_ch, time_embed_dim, num_layers=layers_per_block,
                                           resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                                           attn_num_head_channels=attention_head_dim[i],
                                           cross_attention_dim=cross_attention_dim,
                                           downsample_padding=downsample_padding, add_downsample=not final)
            elif t == "DownBlock3D":
                blk = DownBlock3D(in_ch, out_ch, time_embed_dim, num_layers=layers_per_block, resnet_eps=norm_eps,
                                  resnet_groups=norm_num_groups, add_downsample=not final,
                                  downsample_padding=downsample_padding)
            else:
                raise ValueError(f"{t} does not exist.")
            self.down_blocks.append(blk)
        self.mid_block = UNetMidBlock3DCrossAttn(boc[-1], time_embed_dim, resnet_eps=norm_eps,
                                                 resnet_groups=norm_num_groups,
                                                 attn_num_head_channels=attention_head_dim[-1],
                                                 output_scale_factor=mid_block_scale_factor,
                                                 cross_attention_dim=cross_attention_dim)
        self.num_upsamplers = 0
        rboc = list(reversed(boc))
        rhd = list(reversed(attention_head_dim))
        out_ch = rboc[0]
        for i, t in enumerate(up_block_types):
            final = i == len(boc) - 1
            prev, out_ch = out_ch, rboc[i]
            in_ch = rboc[min(i + 1, len(boc) - 1)]
            if not final:
                self.num_upsamplers += 1
            if t == "UpBlock3D":
                blk = UpBlock3D(in_ch, prev, out_ch, time_embed_dim, num_layers=layers_per_block + 1,
                                resnet_eps=norm_eps, resnet_groups=norm_num_groups, add_upsample=not final)
            elif t == "CrossAttnUpBlock3D":
                blk = CrossAttnUpBlock3D(in_ch, out_ch, prev, time_embed_dim, num_layers=layers_per_block + 1,
                                         resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                                         attn_num_head_channels=rhd[i], cross_attention_dim=cross_attention_dim,
                                         add_upsample=not final)
            else:
                raise ValueError(f"{t} does not exist.")
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, kernel_size=3, padding=1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def _set_gradient_checkpointing(self, value=False):
        self.gradient_checkpointing = value
        self.mid_block.gradient_checkpointing = value
        for m in list(self.down_blocks) + list(self.up_blocks):
            m.gradient_checkpointing = value

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict=True):
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])
        upsample_size = None
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(sample.device)
        num_frames = sample.shape[2]
        timesteps = timesteps.expand(sample.shape[0])
        t_emb = self.time_proj(timesteps).to(dtype=self.dtype)
        emb = self.time_embedding(t_emb, timestep_cond)
        emb = emb.repeat_interleave(repeats=num_frames, dim=0)
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(repeats=num_frames, dim=0)
        sample = sample.permute(0, 2, 1, 3, 4).reshape((sample.shape[0] * num_frames, -1) + sample.shape[3:])
        sample = self.conv_in(sample)
        if num_frames > 1:
            sample = self.transformer_in(sample, num_frames=num_frames).sample
        res_samples = (sample,)
        for blk in self.down_blocks:
            if getattr(blk, "has_cross_attention", False):
                sample, res = blk(hidden_states=sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                  num_frames=num_frames)
            else:
                sample, res = blk(hidden_states=sample, temb=emb, num_frames=num_frames)
            res_samples += res
        if down_block_additional_residuals is not None:
            res_samples = tuple(r + a for r, a in zip(res_samples, down_block_additional_residuals))
        sample = self.mid_block(sample, emb, encoder_hidden_states=encoder_hidden_states, num_frames=num_frames)
        if mid_block_additional_residual is not None:
            sample = sample + mid_block_additional_residual
        for i, blk in enumerate(self.up_blocks):
            final = i == len(self.up_blocks) - 1
            n = len(blk.resnets)
            res, res_samples = res_samples[-n:], res_samples[:-n]
            if not final and forward_upsample_size:
                upsample_size = res_samples[-1].shape[2:]
            if getattr(blk, "has_cross_attention", False):
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res,
                             encoder_hidden_states=encoder_hidden_states, upsample_size=upsample_size,
                             num_frames=num_frames)
            else:
                sample = blk(hidden_states=sample, temb=emb, res_hidden_states_tuple=res,
                             upsample_size=upsample_size, num_frames=num_frames)
        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        sample = sample[None, :].reshape((-1, num_frames) + sample.shape[1:]).permute(0, 2, 1, 3, 4)
        if not return_dict:
            return (sample,)
        return _Out(sample=sample)
