"""General ``skip()`` encoder-decoder (``models/unet/skip.py:4-102``) for the architectures the fused HIP generator
does not cover.

``define_G`` -- the only generator of the Splice hot path -- always builds the default ``skip()`` (5 scales, 3x3 /
1x1 filters, zero padding), and that one runs on the hand-written engine (``splice_amd/networks.py``).  The feature
inversion experiment (``inversion.py:21-25``) asks for a different net: 6 scales, 7/7/5/5/3/3 filters, reflection
padding, 32 noise input channels.  That experiment is outside the hot path; its
generator is assembled here from stock PyTorch-ROCm modules so that ``inversion.py`` runs end to end on the HIP
ViT extractor, which is where its time goes (one ViT forward + backward per iteration).

Structure per scale i (input x_i, ``models/unet/skip.py:46-99``)::

    skip_i  = act(bn(conv_{k_skip}(x_i)))                                  (if num_channels_skip[i] > 0)
    deep_i  = act(bn(conv_{k_down}(act(bn(conv_{k_down, stride 2}(x_i))))))   -> x_{i+1}
    up_i    = upsample_x2( scale_{i+1}(x_{i+1})  or  deep_i at the last scale )
    cat_i   = bn(concat(skip_i, up_i))         (centre-cropped to the smaller size, models/unet/common.py:24-37)
    out_i   = act(bn(conv_{k_up}(cat_i)));  out_i = act(bn(conv_1x1(out_i)))  (if need1x1_up)

and a final ``conv_1x1 -> sigmoid|tanh`` on ``out_0``.  ``downsample_mode`` other than ``'stride'`` and activations
other than LeakyReLU(0.2) are not implemented.
"""
import torch
import torch.nn as nn


def _conv(cin, cout, k, stride=1, bias=True, pad='zero'):
    p = (k - 1) // 2
    layers = []
    if pad == 'reflection':
        if p:
            layers.append(nn.ReflectionPad2d(p))
        p = 0
    elif pad != 'zero':
        raise NotImplementedError(f"pad mode {pad!r}")
    layers.append(nn.Conv2d(cin, cout, k, stride, padding=p, bias=bias))
    return nn.Sequential(*layers)


def _unit(cin, cout, k, stride=1, bias=True, pad='zero'):
    return nn.Sequential(_conv(cin, cout, k, stride, bias, pad), nn.BatchNorm2d(cout), nn.LeakyReLU(0.2))


def _center_crop(x, h, w):
    dh, dw = (x.shape[2] - h) // 2, (x.shape[3] - w) // 2
    return x[:, :, dh:dh + h, dw:dw + w]


class _Scale(nn.Module):
    def __init__(self, cin, i, down, up, skipc, k_down, k_up, k_skip, bias, pad, upsample_mode, need1x1_up):
        super().__init__()
        last = i == len(down) - 1
        self.skip = _unit(cin, skipc[i], k_skip, 1, bias, pad) if skipc[i] else None
        self.down = nn.Sequential(_unit(cin, down[i], k_down[i], 2, bias, pad), _unit(down[i], down[i], k_down[i], 1, bias, pad))
        self.inner = None if last else _Scale(down[i], i + 1, down, up, skipc, k_down, k_up, k_skip, bias, pad, upsample_mode, need1x1_up)
        kch = down[i] if last else up[i + 1]
        self.up = nn.Upsample(scale_factor=2, mode=upsample_mode[i], **({"align_corners": False} if upsample_mode[i] == "bilinear" else {}))
        self.cat_bn = nn.BatchNorm2d(skipc[i] + kch)
        layers = [_unit(skipc[i] + kch, up[i], k_up[i], 1, bias, pad)]
        if need1x1_up:
            layers.append(_unit(up[i], up[i], 1, 1, bias, pad))
        self.out = nn.Sequential(*layers)

    def forward(self, x):
        d = self.down(x)
        if self.inner is not None:
            d = self.inner(d)
        d = self.up(d)
        if self.skip is not None:
            s = self.skip(x)
            h, w = min(s.shape[2], d.shape[2]), min(s.shape[3], d.shape[3])
            d = torch.cat([_center_crop(s, h, w), _center_crop(d, h, w)], dim=1)
        return self.out(self.cat_bn(d))


class GeneralSkip(nn.Module):
    def __init__(self, num_input_channels=3, num_output_channels=3, num_channels_down=(16, 32, 64, 128, 128),
                 num_channels_up=(16, 32, 64, 128, 128), num_channels_skip=(4, 4, 4, 4, 4), filter_size_down=3, filter_size_up=3,
                 filter_skip_size=1, need_sigmoid=True, need_tanh=False, need_bias=True, pad='zero', upsample_mode='bilinear',
                 downsample_mode='stride', act_fun='LeakyReLU', need1x1_up=True):
        super().__init__()
        n = len(num_channels_down)
        if not (n == len(num_channels_up) == len(num_channels_skip)):
            raise ValueError("num_channels_down / _up / _skip must have one entry per scale")
        ds = downsample_mode if isinstance(downsample_mode, (list, tuple)) else [downsample_mode] * n
        if any(m != 'stride' for m in ds) or act_fun != 'LeakyReLU':
            raise NotImplementedError("GeneralSkip implements downsample_mode='stride' and act_fun='LeakyReLU'")
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * n
        self.body = _Scale(num_input_channels, 0, list(num_channels_down), list(num_channels_up), list(num_channels_skip),
                           as_list(filter_size_down), as_list(filter_size_up), filter_skip_size, need_bias, pad,
                           as_list(upsample_mode), need1x1_up)
        self.head = _conv(num_channels_up[0], num_output_channels, 1, 1, need_bias, pad)
        self.final = nn.Sigmoid() if need_sigmoid else (nn.Tanh() if need_tanh else nn.Identity())

    def forward(self, x):
        return self.final(self.head(self.body(x)))
