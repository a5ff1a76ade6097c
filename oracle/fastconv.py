"""Timed-baseline helper (ORACLE — test infrastructure): PyTorch's CPU `conv3d` takes a slow generic path
(3-17 GFLOP/s measured) for the (k,1,1) temporal kernels this model uses.  Inside
`fast_temporal_conv3d()` every `nn.Conv3d` with a (k,1,1) kernel evaluates the SAME convolution as a (k,1)
`conv2d` over (frames, H*W) — identical arithmetic up to fp32 summation order — so that the CPU baseline is not
handicapped by a library corner case.  Checked against the plain path in tests/test_oracle_pins.py."""
import contextlib

import torch
import torch.nn.functional as F
from torch import nn


def _conv3d_as_conv2d(self, x, weight, bias):
    k = self.kernel_size
    if (k[1], k[2]) != (1, 1) or self.stride != (1, 1, 1) or self.dilation != (1, 1, 1) or self.groups != 1 \
            or self.padding_mode != "zeros" or isinstance(self.padding, str) or x.dim() != 5:
        return self._t2v_orig_conv_forward(x, weight, bias)
    b, c, f, h, w = x.shape
    y = F.conv2d(x.reshape(b, c, f, h * w), weight.reshape(weight.shape[0], weight.shape[1], k[0], 1), bias,
                 padding=(self.padding[0], 0))
    return y.reshape(b, weight.shape[0], y.shape[2], h, w)


@contextlib.contextmanager
def fast_temporal_conv3d():
    orig = nn.Conv3d._conv_forward
    nn.Conv3d._t2v_orig_conv_forward = orig
    nn.Conv3d._conv_forward = _conv3d_as_conv2d
    try:
        yield
    finally:
        nn.Conv3d._conv_forward = orig
        del nn.Conv3d._t2v_orig_conv_forward
