"""CPU fp32 restatement of `AutoencoderKL.encode` / `.decode` (ORACLE — test infrastructure).

Call site in the reference: `train.py:339-347` (`tensor_to_vae_latent`):
`vae.encode(t).latent_dist.sample()` on `(B*F,3,H,W)`, then `* 0.18215`.
The encoder itself is un-vendored diffusers code; semantics restated from
SURVEY.md Appendix A.7 (SD-VAE: block_out (128,256,512,512), 2 resnets per
block, GroupNorm(32, eps 1e-6), SiLU, single-head mid attention d=512).
Attribute names follow diffusers (`encoder.down_blocks.{i}.resnets.{j}`,
`encoder.mid_block.attentions.0.to_q`, `quant_conv`).

Decoder (call sites `inference.py:125-140` `pipe.vae.decode(latents / scaling_factor).sample`, and the
pipeline's `decode_latents` behind `train.py:918-943`): post_quant_conv (1x1) -> conv_in (4 -> 512) -> mid block (resnet,
single-head attention, resnet) -> 4 UpDecoderBlock2D over the reversed channel list (512, 512, 256, 128), layers_per_block + 1
= 3 resnets each and a nearest-2x upsample + 3x3 conv on all but the last -> GroupNorm(32, eps 1e-6) -> SiLU -> conv_out
(128 -> 3).  Attribute names follow diffusers (`decoder.up_blocks.{i}.resnets.{j}`, `decoder.up_blocks.{i}.upsamplers.0.conv`,
`post_quant_conv`).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .unet3d import Downsample2D, ResnetBlock2D, Upsample2D


class VaeAttention(nn.Module):
    """Single-head spatial self-attention of the VAE mid block (GroupNorm -> q,k,v -> softmax -> out, residual)."""

    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        n, c, h, w = x.shape
        res = x
        t = self.group_norm(x).view(n, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.matmul(q, k.transpose(1, 2)) * (c ** -0.5), dim=-1)
        o = self.to_out[0](torch.matmul(p, v))
        return o.transpose(1, 2).reshape(n, c, h, w) + res


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, num_layers=2, add_downsample=True, eps=1e-6, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None, eps=eps,
                          groups=groups) for i in range(num_layers)])
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, use_conv=True, out_channels=cout, padding=0,
                                                            name="op")])

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
        return x


class VaeMidBlock(nn.Module):
    def __init__(self, c, eps=1e-6, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=c, out_channels=c, temb_channels=None, eps=eps, groups=groups)
            for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups, eps)])

    def forward(self, x):
        x = self.resnets[0](x, None)
        x = self.attentions[0](x)
        return self.resnets[1](x, None)


class Encoder(nn.Module):
    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 groups=32):
        super().__init__()
        boc = block_out_channels
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([])
        cout = boc[0]
        for i, c in enumerate(boc):
            cin, cout = cout, c
            self.down_blocks.append(DownEncoderBlock2D(cin, cout, layers_per_block, add_downsample=i != len(boc) - 1,
                                                       groups=groups))
        self.mid_block = VaeMidBlock(boc[-1], groups=groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class AutoencoderKLEncoder(nn.Module):
    """Encoder half of AutoencoderKL: `encode_moments(x)` -> (mean, logvar); `sample(...)` applies the
    diagonal-Gaussian reparameterisation with caller-supplied noise (host-injected randomness)."""

    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 groups=32):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def encode_moments(self, x):
        m = self.quant_conv(self.encoder(x))
        mean, logvar = torch.chunk(m, 2, dim=1)
        return mean, torch.clamp(logvar, -30.0, 20.0)

    def encode_sample(self, x, eps):
        mean, logvar = self.encode_moments(x)
        return mean + torch.exp(0.5 * logvar) * eps


def tensor_to_vae_latent(pixel_values, vae, eps):
    """train.py:339-347 with the posterior noise `eps` (B*F,4,h,w) injected by the caller."""
    b, f = pixel_values.shape[:2]
    t = pixel_values.reshape((b * f,) + pixel_values.shape[2:])
    lat = vae.encode_sample(t, eps)
    lat = lat.reshape((b, f) + lat.shape[1:]).permute(0, 2, 1, 3, 4)
    return lat * 0.18215


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, num_layers=3, add_upsample=True, eps=1e-6, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None, eps=eps,
                          groups=groups) for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, use_conv=True, out_channels=cout)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x)
        return x


class Decoder(nn.Module):
    def __init__(self, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 groups=32):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = VaeMidBlock(rev[0], groups=groups)
        self.up_blocks = nn.ModuleList([])
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            self.up_blocks.append(UpDecoderBlock2D(cin, cout, layers_per_block + 1, add_upsample=i != len(rev) - 1,
                                                   groups=groups))
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class AutoencoderKLDecoder(nn.Module):
    """Decoder half of AutoencoderKL: `decode(z)` = decoder(post_quant_conv(z)) (the `.sample` of diffusers' DecoderOutput)."""

    def __init__(self, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 groups=32):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = Decoder(out_channels, latent_channels, block_out_channels, layers_per_block, groups)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def decode_latents(latents, vae, batch_size=8, scaling_factor=0.18215):
    """inference.py:125-140: `(B,4,F,h,w)` latents -> `(B,3,F,H,W)` fp32 frames, decoded `batch_size` frames at a time."""
    b, c, f, h, w = latents.shape
    x = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    out = [vae.decode(x[i:i + batch_size] / scaling_factor) for i in range(0, b * f, batch_size)]
    px = torch.cat(out)
    return px.reshape(b, f, px.shape[1], px.shape[2], px.shape[3]).permute(0, 2, 1, 3, 4).float()
