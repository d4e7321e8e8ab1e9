"""Functional restatement of the reference generator (test infrastructure only).

``define_G`` -> ``skip()`` with its defaults (``models/networks.py:56-58``,
``models/unet/skip.py:4-11``): 5 scales, down/up channels [16,32,64,128,128], 4 skip
channels, 3x3 filters (1x1 skip and post-up), zero padding, bias, LeakyReLU(0.2),
train-mode BatchNorm2d (batch statistics, biased variance, eps 1e-5), bilinear x2
up-sampling (align_corners=False), centre-cropping Concat, final 1x1 conv + Sigmoid.

Parameters live in a flat dict keyed by the reference's ``state_dict`` names (the
``nn.Module.add`` monkey-patch of ``models/unet/common.py:6-9`` starts numbering at
'1'), so weights can be exchanged with ``define_G()`` verbatim.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

DOWN = [16, 32, 64, 128, 128]
UP = [16, 32, 64, 128, 128]
SKIP = [4, 4, 4, 4, 4]
N_SCALES = 5
BN_EPS = 1e-5
LRELU = 0.2


def scale_prefix(i):
    return "1.1.7." * i


def param_specs():
    """Ordered (name, shape, kind) for all 112 parameter tensors, in the order
    ``netG.parameters()`` yields them (registration order, skip.py:46-99)."""
    specs = []

    def conv(name, cout, cin, k):
        specs.append((name + ".weight", (cout, cin, k, k), "conv_w"))
        specs.append((name + ".bias", (cout,), "conv_b"))

    def bn(name, c):
        specs.append((name + ".weight", (c,), "bn_w"))
        specs.append((name + ".bias", (c,), "bn_b"))

    def scale(i, cin):
        p = scale_prefix(i)
        conv(p + "1.0.1.0", SKIP[i], cin, 1)
        bn(p + "1.0.2", SKIP[i])
        conv(p + "1.1.1.0", DOWN[i], cin, 3)
        bn(p + "1.1.2", DOWN[i])
        conv(p + "1.1.4.0", DOWN[i], DOWN[i], 3)
        bn(p + "1.1.5", DOWN[i])
        if i < N_SCALES - 1:
            scale(i + 1, DOWN[i])
            k = UP[i + 1]
        else:
            k = DOWN[i]
        bn(p + "2", SKIP[i] + k)
        conv(p + "3.0", UP[i], SKIP[i] + k, 3)
        bn(p + "4", UP[i])
        conv(p + "6.0", UP[i], UP[i], 1)
        bn(p + "7", UP[i])

    scale(0, 3)
    conv("9.0", 3, UP[0], 1)
    return specs


def init_params(init_type="xavier", init_gain=0.02, generator=None):
    """``models/networks.py:24-47``: conv W ~ xavier_normal(gain), conv b = 0,
    BN gamma ~ N(1, gain), BN beta = 0."""
    assert init_type == "xavier"
    params = OrderedDict()
    for name, shape, kind in param_specs():
        if kind == "conv_w":
            cout, cin, k, _ = shape
            std = init_gain * math.sqrt(2.0 / ((cin + cout) * k * k))
            params[name] = torch.randn(shape, generator=generator) * std
        elif kind == "bn_w":
            params[name] = 1.0 + init_gain * torch.randn(shape, generator=generator)
        else:
            params[name] = torch.zeros(shape)
    return params


def _bn_act(x, params, name, act=True):
    y = F.batch_norm(x, None, None, params[name + ".weight"], params[name + ".bias"],
                     training=True, momentum=0.1, eps=BN_EPS)
    return F.leaky_relu(y, LRELU) if act else y


def _conv(x, params, name, stride=1):
    w = params[name + ".weight"]
    return F.conv2d(x, w, params[name + ".bias"], stride=stride, padding=(w.shape[-1] - 1) // 2)


def _concat_crop(a, b):
    """``models/unet/common.py:19-37``: centre-crop both to the smaller H and W."""
    th, tw = min(a.shape[2], b.shape[2]), min(a.shape[3], b.shape[3])
    outs = []
    for t in (a, b):
        d2, d3 = (t.shape[2] - th) // 2, (t.shape[3] - tw) // 2
        outs.append(t[:, :, d2:d2 + th, d3:d3 + tw])
    return torch.cat(outs, dim=1)


def _scale_forward(x, params, i):
    p = scale_prefix(i)
    s = _bn_act(_conv(x, params, p + "1.0.1.0"), params, p + "1.0.2")
    d = _bn_act(_conv(x, params, p + "1.1.1.0", stride=2), params, p + "1.1.2")
    d = _bn_act(_conv(d, params, p + "1.1.4.0"), params, p + "1.1.5")
    if i < N_SCALES - 1:
        d = _scale_forward(d, params, i + 1)
    d = F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=False)
    y = _bn_act(_concat_crop(s, d), params, p + "2", act=False)
    y = _bn_act(_conv(y, params, p + "3.0"), params, p + "4")
    return _bn_act(_conv(y, params, p + "6.0"), params, p + "7")


def forward(params, x):
    """x ``[N,3,H,W]`` in [0,1] -> same shape in (0,1).  The reference always calls the
    net with N=1 (per-crop); BN statistics are per call."""
    return torch.sigmoid(_conv(_scale_forward(x, params, 0), params, "9.0"))
