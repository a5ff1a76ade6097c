"""Native VAE: encoder for `tensor_to_vae_latent` (reference: `train.py:339-347`, `handle_cache_latents :266-314`) and decoder
for the sampling side (`inference.py:125-140`, the pipeline's `decode_latents` behind `train.py:918-943`).

`AutoencoderKL.encode(x).latent_dist.sample()` of diffusers restated over the HIP kernels: conv_in, 4
DownEncoderBlock2D (2 resnets each, stride-2 pad-(0,1,0,1) downsample on the first three), mid block
(resnet, single-head d=512 attention, resnet), GroupNorm+SiLU, conv_out, quant_conv, diagonal-Gaussian sample
(SURVEY.md Appendix A.7).  Attribute names/state-dict keys follow diffusers (`encoder.down_blocks.i.resnets.j`, …).
The d=512 attention is computed per frame with the batched GEMM (scores -> row softmax -> P V); it is <1 % of the
encoder FLOPs.  Forward only: the VAE is frozen (`train.py:543`).
Decoder: post_quant_conv -> conv_in -> mid block -> 4 UpDecoderBlock2D over (512, 512, 256, 128) with 3 resnets each and a
nearest-2x upsample folded into the following 3x3 conv's gather (no upsampled tensor is materialised) -> GroupNorm+SiLU -> conv_out.
"""
import torch
from torch import nn

from .. import functional as F
from .. import native as nv
from ..functional import ConvCfg
from .leaves import Downsample2D, ResnetBlock2D, Tok, Upsample2D, run_layer
from .modeling_utils import ModelMixinLite

BF16 = torch.bfloat16


class VaeAttention(nn.Module):
    def __init__(self, channels, groups=32, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        n, S, Cc = x.n, x.h * x.w, x.C
        a = F.group_norm(x.m, self.group_norm.weight, self.group_norm.bias, self.group_norm.num_groups,
                         self.group_norm.eps, False, n)
        q, k, v = run_layer(self.to_q, a), run_layer(self.to_k, a), run_layer(self.to_v, a)
        Sp = F.ceil8(S)
        scores = torch.empty(n * S, Sp, dtype=BF16, device=a.device)
        F.launch_gemm(M=S, N=S, K=Cc, A=q.data_ptr(), lda=Cc, B=k.data_ptr(), ldb=Cc, D=scores.data_ptr(), ldd=Sp,
                      alpha=float(Cc) ** -0.5, batch=n, strideA=S * Cc, strideB=S * Cc, strideD=S * Sp)
        if Sp != S:
            raise RuntimeError("t2v_amd: VAE attention needs H*W/64 to be a multiple of 8")
        nv.call("t2v_softmax_rows", scores.data_ptr(), Sp, scores.data_ptr(), Sp, n * S, S, nv.stream())
        o = torch.empty(n * S, Cc, dtype=BF16, device=a.device)
        F.launch_gemm(M=S, N=Cc, K=S, A=scores.data_ptr(), lda=Sp, B=v.data_ptr(), ldb=Cc, b_trans=1, D=o.data_ptr(),
                      ldd=Cc, batch=n, strideA=S * Sp, strideB=S * Cc, strideD=S * Cc)
        return Tok(run_layer(self.to_out[0], o, residual=x.m), x.n, x.h, x.w)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, num_layers=2, add_downsample=True, eps=1e-6, groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout, temb_channels=None, eps=eps,
                          groups=groups) for i in range(num_layers)])
        self.downsamplers = None
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(cout, use_conv=True, out_channels=cout, padding=0, name="op")])

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
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=c, out_channels=c, temb_channels=None, eps=eps,
                                                    groups=groups) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups, eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, None)), None)


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
        x = Tok(run_layer(self.conv_in, x.m, ConvCfg.conv2d(x.n, x.h, x.w, 3, 1, 1)), x.n, x.h, x.w)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        a = F.group_norm(x.m, self.conv_norm_out.weight, self.conv_norm_out.bias, self.conv_norm_out.num_groups,
                         self.conv_norm_out.eps, True, x.n)
        return Tok(run_layer(self.conv_out, a, ConvCfg.conv2d(x.n, x.h, x.w, 3, 1, 1)), x.n, x.h, x.w)


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

    def forward(self, x):
        x = Tok(run_layer(self.conv_in, x.m, ConvCfg.conv2d(x.n, x.h, x.w, 3, 1, 1)), x.n, x.h, x.w)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        a = F.group_norm(x.m, self.conv_norm_out.weight, self.conv_norm_out.bias, self.conv_norm_out.num_groups,
                         self.conv_norm_out.eps, True, x.n)
        return Tok(run_layer(self.conv_out, a, ConvCfg.conv2d(x.n, x.h, x.w, 3, 1, 1)), x.n, x.h, x.w)


class DiagonalGaussianDistribution:
    def __init__(self, mean, logvar):
        self.mean = mean
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, eps=None):
        if eps is None:
            eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps.to(self.mean.device, self.mean.dtype)

    def mode(self):
        return self.mean


class _EncOut:
    def __init__(self, dist):
        self.latent_dist = dist


class _DecOut:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL(nn.Module, ModelMixinLite):
    """The SD VAE.  `encode(x).latent_dist.sample()` as in train.py:343; `decode(z).sample` as in inference.py:134.
    `with_decoder=False` (the constructor default: the train step never decodes, 49 M parameters less to hold) builds the encoder
    half only; `from_pretrained` builds the decoder when the checkpoint carries one (override with `with_decoder=`)."""

    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, with_decoder=False, **_):
        super().__init__()
        self.register_to_config(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                norm_num_groups=norm_num_groups, scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, tuple(block_out_channels), layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.decoder = self.post_quant_conv = None
        if with_decoder:
            self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
            self.decoder = Decoder(out_channels, latent_channels, tuple(block_out_channels), layers_per_block, norm_num_groups)
        self.use_slicing = False

    @classmethod
    def _ctor_kwargs_from_checkpoint(cls, sd):
        return {"with_decoder": any(k.startswith("decoder.") for k in sd)}

    def _convert_checkpoint_keys(self, sd):
        """A stock diffusers `vae/diffusion_pytorch_model.*` carries both halves and — in checkpoints of the ModelScope era —
        the deprecated attention names `query/key/value/proj_attn`.  Rename the latter; drop the decoder half (`decoder.*`,
        `post_quant_conv.*`) when this instance was built without it (`with_decoder=False`: the train step only encodes,
        train.py:339-347).  The load itself stays strict on what remains."""
        ren = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}
        out = {}
        for k, v in sd.items():
            if self.decoder is None and k.startswith(("decoder.", "post_quant_conv.")):
                continue
            for a, b in ren.items():
                if a in k and ".attentions." in k:
                    k = k.replace(a, b)
            out[k] = v
        return out

    def enable_slicing(self):   # train.py:280,678 — batching knob only, results unchanged
        self.use_slicing = True

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        if not x.is_cuda:
            raise RuntimeError("t2v_amd.AutoencoderKL runs on a ROCm device only (the CPU restatement is oracle/vae.py)")
        lc = self.config.latent_channels
        t = self.encoder(Tok.from_nchw(x))
        m = run_layer(self.quant_conv, t.m, ConvCfg.conv2d(t.n, t.h, t.w, 1, 1, 0))
        mom = Tok(m, t.n, t.h, t.w).to_nchw(2 * lc, torch.float32)
        dist = DiagonalGaussianDistribution(mom[:, :lc], mom[:, lc:])
        return _EncOut(dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """`(N,4,h,w)` latents (already divided by the scaling factor) -> `.sample` `(N,3,8h,8w)` fp32."""
        if self.decoder is None:
            raise RuntimeError("t2v_amd.AutoencoderKL was built without its decoder half (with_decoder=False)")
        if not z.is_cuda:
            raise RuntimeError("t2v_amd.AutoencoderKL runs on a ROCm device only (the CPU restatement is oracle/vae.py)")
        t = Tok.from_nchw(z)
        m = run_layer(self.post_quant_conv, t.m, ConvCfg.conv2d(t.n, t.h, t.w, 1, 1, 0))
        y = self.decoder(Tok(m, t.n, t.h, t.w))
        px = y.to_nchw(self.config.out_channels, torch.float32)
        return _DecOut(px) if return_dict else (px,)


def decode_latents(latents, vae, batch_size=8):
    """inference.py:125-140 (`decode`): `(B,4,F,h,w)` latents -> `(B,3,F,H,W)` fp32 frames in [-1,1], `batch_size` frames per
    VAE call, `/ scaling_factor` first."""
    b, c, f, h, w = latents.shape
    x = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    sf = vae.config.scaling_factor
    px = torch.cat([vae.decode(x[i:i + batch_size] / sf).sample for i in range(0, b * f, batch_size)])
    return px.reshape(b, f, px.shape[1], px.shape[2], px.shape[3]).permute(0, 2, 1, 3, 4).float()


def tensor_to_vae_latent(t, vae, eps=None):
    """train.py:339-347; `eps` optionally injects the posterior noise (host-seeded parity runs)."""
    b, f = t.shape[:2]
    x = t.reshape((b * f,) + tuple(t.shape[2:]))
    lat = vae.encode(x).latent_dist.sample(eps=eps)
    lat = lat.reshape((b, f) + tuple(lat.shape[1:])).permute(0, 2, 1, 3, 4)
    return lat * 0.18215
