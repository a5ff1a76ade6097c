"""Counter-based dropout masks (ORACLE — test infrastructure): CPU restatement of the keep decision the HIP kernels use.

The reference draws its dropout masks from the device RNG (`nn.Dropout` in TemporalConvLayer, models/unet_3d_blocks.py:312…;
LoRA branch, utils/lora.py:49,119), which cannot be reproduced across devices.  SURVEY 8(d) asks for a mask that is
"identically defined on CPU and GPU".  Protocol v2 (round 4; csrc/common.h `drop_key` / `drop_quad`): elements are decided
four at a time.  With idx = row * row_width + column of the token matrix the mask is applied to (row_width % 8 == 0),
quad q = idx >> 2, element e = idx & 3:

    z  = splitmix64-finaliser(seed) ; s0 = low 32 bits, s1 = high 32 bits                       (once per launch)
    a  = fmix32((q mod 2^32) ^ s0 ^ ((q >> 32) * 0x9E3779B1))          murmur3 finaliser, 32-bit arithmetic
    b  = (a ^ s1) * 0x9E3779B1 ; b ^= b >> 15 ; b *= 0x85EBCA77 ; b ^= b >> 13
    field(e) = a & 0xffff, a >> 16, b & 0xffff, b >> 16      for e = 0, 1, 2, 3
    keep  iff  field(e) >= round(p * 65536)

(the first protocol hashed every element with two 64-bit multiplies — the mask arithmetic bound the kernels that carry it).
"""
import numpy as np
import torch

_M64 = (1 << 64) - 1
_M32 = (1 << 32) - 1


def keep_mask(seed, rows, cols, p):
    """bool [rows, cols]: True where element (row, col) of a [rows, cols] token matrix is kept (cols % 4 == 0)."""
    if cols % 4:
        raise ValueError("mask rows are whole quads: the column count must be a multiple of 4")
    z = _mix64(int(seed) & _M64)
    s0, s1 = np.uint32(z & _M32), np.uint32(z >> 32)
    thr = np.uint32(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)))
    nq = rows * cols // 4
    q = np.arange(nq, dtype=np.uint64)
    with np.errstate(over="ignore"):
        a = (q & np.uint64(_M32)).astype(np.uint32) ^ s0 ^ ((q >> np.uint64(32)).astype(np.uint32) * np.uint32(0x9E3779B1))
        a ^= a >> np.uint32(16)
        a *= np.uint32(0x85EBCA6B)
        a ^= a >> np.uint32(13)
        a *= np.uint32(0xC2B2AE35)
        a ^= a >> np.uint32(16)
        b = (a ^ s1) * np.uint32(0x9E3779B1)
        b ^= b >> np.uint32(15)
        b *= np.uint32(0x85EBCA77)
        b ^= b >> np.uint32(13)
    f = np.stack([a & np.uint32(0xFFFF), a >> np.uint32(16), b & np.uint32(0xFFFF), b >> np.uint32(16)], axis=1)
    return torch.from_numpy((f >= thr).reshape(rows, cols))


def _mix64(z):
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def effective_seed(seed, epoch):
    """Seed of a launch issued while a dropout epoch is registered (`t2v_set_dropout_epoch`, csrc/common.h `eff_seed`):
    seed XOR splitmix64(epoch + 0x9E3779B97F4A7C15); epoch = the value of the device counter when the kernel runs.  The epoch
    is hashed BEFORE it meets the seed — added on the index lattice (seed + epoch * G, the first version) it made the mask of
    step e+1 the mask of step e shifted by one element."""
    return (int(seed) ^ _mix64(int(epoch) + 0x9E3779B97F4A7C15)) & _M64


def apply(x, seed, p):
    """Inverted dropout of a [rows, cols] matrix with the protocol's mask."""
    m = keep_mask(seed, x.shape[0], x.shape[1], p).to(x.dtype)
    return x * m / (1.0 - p)


# ----------------------------------------------------------------------------------------------------------------------
# Model-level protocol: which seed and which element index every dropout SITE of the UNet uses in the native path, restated for
# the oracle so that the reference's default train mode (LoRA dropout 0.1, utils/lora.py:35,89; TemporalConvLayer dropout 0.1,
# models/unet_3d_blocks.py:312) can be compared mask for mask.
def site_seed(base, name, step=0):
    """models/leaves.py::_seed_for — seed of the dropout site `name` (qualified module name [+ suffix])."""
    import zlib
    return (int(base) * 1000003 + zlib.crc32(name.encode()) * 97 + int(step) * 7919) & 0xFFFFFFFFFFFF


class ProtocolDropout(torch.nn.Module):
    """Stands in for one nn.Dropout of the oracle UNet.  The native path applies its masks on channels-last token matrices with
    rows ordered (pass, batch, frame, y, x) — the two UNet passes of train.py:814-834 are one stacked forward there — and pads the
    column count to a multiple of 8; the element index is row * padded_columns + column.  `kind` names the layout of the tensor
    this site sees in the ORACLE:
      cl5  [B, C, F, H, W]            cl4  [B*F, C, H, W]          tok3 [B*F, S, C]   (spatial tokens)
      tmp3 [B*H*W, F, C] (temporal transformer: native rows stay (b, f, pixel))       row2 [B, C]
      rowf [B*F, C]      (time_emb_proj: the reference repeats the time embedding per frame, models/unet_3d_condition.py:400;
                          the native path keeps it per video — a row-bias of the conv epilogue — so every frame shares the mask)
      txt3 [B*F, 77, C]  (text keys / values: the native path projects the text states ONCE per video, every frame shares it)"""

    def __init__(self, p, seed, kind, ctx):
        super().__init__()
        self.p, self.seed, self.kind, self.ctx = float(p), int(seed), kind, ctx

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        c = self.ctx
        K, k, Fr = c["passes"], c["k"], c["frames"]
        seed = effective_seed(self.seed, c["epoch"]) if c.get("epoch") is not None else self.seed
        C_ = x.shape[1] if self.kind in ("cl5", "cl4") else x.shape[-1]
        Cp = (C_ + 7) // 8 * 8
        if self.kind == "cl5":
            B, _, F_, H, W = x.shape
            m = keep_mask(seed, K * B * F_ * H * W, Cp, self.p).view(K * B, F_, H, W, Cp)[k * B:(k + 1) * B, ..., :C_]
            m = m.permute(0, 4, 1, 2, 3)
        elif self.kind == "cl4":
            N, _, H, W = x.shape
            m = keep_mask(seed, K * N * H * W, Cp, self.p).view(K * N, H, W, Cp)[k * N:(k + 1) * N, ..., :C_].permute(0, 3, 1, 2)
        elif self.kind == "tok3":
            N, S, _ = x.shape
            m = keep_mask(seed, K * N * S, Cp, self.p).view(K * N, S, Cp)[k * N:(k + 1) * N, :, :C_]
        elif self.kind == "tmp3":
            NB, F_, _ = x.shape                       # NB = B * H*W
            B = c["batch"]
            HW = NB // B
            m = keep_mask(seed, K * B * F_ * HW, Cp, self.p).view(K * B, F_, HW, Cp)[k * B:(k + 1) * B, ..., :C_]
            m = m.permute(0, 2, 1, 3).reshape(NB, F_, C_)
        elif self.kind == "txt3":
            N, S, _ = x.shape                         # N = B * frames
            B = N // Fr
            m = keep_mask(seed, K * B * S, Cp, self.p).view(K * B, S, Cp)[k * B:(k + 1) * B, :, :C_].repeat_interleave(Fr, dim=0)
        elif self.kind == "row2":
            B = x.shape[0]
            m = keep_mask(seed, K * B, Cp, self.p).view(K * B, Cp)[k * B:(k + 1) * B, :C_]
        elif self.kind == "rowf":                     # [B*F, C]: the native path evaluates the site once per video
            B = x.shape[0] // Fr
            m = keep_mask(seed, K * B, Cp, self.p).view(K * B, Cp)[k * B:(k + 1) * B, :C_].repeat_interleave(Fr, dim=0)
        else:
            raise ValueError(self.kind)
        return x * m.to(x.dtype) / (1.0 - self.p)


def install_protocol(unet, base_seed, step, epoch, batch, frames, passes=2):
    """Replace every active nn.Dropout of the oracle UNet (LoRA wrappers, TemporalConvLayer sequences) by its ProtocolDropout.
    Returns the shared context; a forward-pre-hook on the UNet advances the pass index (the oracle runs the passes one by one)."""
    ctx = dict(passes=passes, k=-1, frames=frames, batch=batch, epoch=epoch)
    unet.register_forward_pre_hook(lambda m, a: ctx.__setitem__("k", (ctx["k"] + 1) % passes))
    nn = torch.nn
    for name, mod in list(unet.named_modules()):
        cls = mod.__class__.__name__
        if cls.startswith("LoraInjected") and isinstance(getattr(mod, "dropout", None), nn.Dropout) and mod.dropout.p > 0:
            if cls == "LoraInjectedConv3d":
                kind = "cl5"
            elif cls == "LoraInjectedConv2d":
                kind = "cl4"
            elif "temp_attentions" in name or name.startswith("transformer_in"):
                kind = "tmp3"
            elif name.endswith("attn2.to_k") or name.endswith("attn2.to_v"):
                kind = "txt3"
            elif name.endswith("time_emb_proj"):
                kind = "rowf"
            elif name.startswith("time_embedding"):
                kind = "row2"
            else:
                kind = "tok3"
            mod.dropout = ProtocolDropout(mod.dropout.p, site_seed(base_seed, name, step), kind, ctx)
        elif cls == "TemporalConvLayer":
            for i, seq in enumerate((mod.conv1, mod.conv2, mod.conv3, mod.conv4)):
                for j, sub in enumerate(seq):
                    if isinstance(sub, nn.Dropout) and sub.p > 0:
                        seq[j] = ProtocolDropout(sub.p, site_seed(base_seed, name + f".conv{i + 1}", step), "cl5", ctx)
    return ctx
