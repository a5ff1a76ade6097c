"""Counter-based dropout masks (ORACLE — test infrastructure): CPU restatement of the keep decision the HIP kernels use.

The reference draws its dropout masks from the device RNG (`nn.Dropout` in TemporalConvLayer, models/unet_3d_blocks.py:312…;
LoRA branch, utils/lora.py:49,119), which cannot be reproduced across devices.  SURVEY 8(d) asks for a mask that is
"identically defined on CPU and GPU": element `idx` of a tensor is kept iff

    z = seed + (idx + 1) * 0x9E3779B97F4A7C15          (mod 2^64)
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9 ; z = (z ^ (z >> 27)) * 0x94D049BB133111EB ; z ^= z >> 31      (splitmix64)
    u = (z >> 40) / 2^24  >=  p

(csrc/common.h `drop_keep`).  idx = row * row_width + column of the token matrix the mask is applied to.
"""
import numpy as np
import torch

_M64 = (1 << 64) - 1


def keep_mask(seed, rows, cols, p):
    """bool [rows, cols]: True where element (row, col) of a [rows, cols] token matrix is kept."""
    idx = np.arange(rows * cols, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u >= np.float32(p)).reshape(rows, cols))


def effective_seed(seed, epoch):
    """Seed of a launch issued while a dropout epoch is registered (`t2v_set_dropout_epoch`, csrc/common.h `eff_seed`):
    seed + epoch * 0x9E3779B97F4A7C15 (mod 2^64); epoch = the value of the device counter when the kernel runs."""
    return (int(seed) + int(epoch) * 0x9E3779B97F4A7C15) & _M64


def apply(x, seed, p):
    """Inverted dropout of a [rows, cols] matrix with the protocol's mask."""
    m = keep_mask(seed, x.shape[0], x.shape[1], p).to(x.dtype)
    return x * m / (1.0 - p)
