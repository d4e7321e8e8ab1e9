"""Device-side counterpart of the reference's PIL augmentation pipelines.

``data/transforms.py:30-41``::

    dino_structure_transforms = [RandomHorizontalFlip(0.5),
                                 RandomApply([ColorJitter(.4, .4, .2, .1)], p=0.5),
                                 RandomApply([GaussianBlur(kernel_size=3)], p=0.2)]
    dino_texture_transforms   = [RandomHorizontalFlip(0.5)]

followed by ``Global_crops`` (``data/transforms.py:7-27``): one random square crop whose side is
``round(U(min_cover*h, h))`` clipped to the width.  The reference runs these on PIL images on the host every
step (``data/Dataset.py:62-70``, serial with the optimisation step); here the image stays on the GPU as a
``[3,H,W]`` float tensor in [0,1] and the same operations run on it (torchvision-0.10's own tensor code path:
``functional_tensor.adjust_*``, ``gaussian_blur``): as HIP kernels when the image is on the GPU
(``splice_augment_structure``, the whole structure pipeline in at most three launches), as the torch ops below
otherwise.  The reference's order is kept: the whole image is augmented, THEN cropped, so ColorJitter's contrast uses
the mean of the whole image.

Random draws are made in the order torchvision makes them (``torch.rand`` for the flips and ``RandomApply``,
``randperm`` + four ``uniform_`` for ColorJitter, ``uniform_`` for the blur sigma, ``np.random.uniform`` for the crop
size, ``torch.randint`` for the crop corner); what is pinned by the tests is the DISTRIBUTION of every parameter and
the arithmetic of every op (against PIL / an independent numpy restatement), not the PIL uint8 rounding.
"""
import math

import numpy as np
import torch

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0.4, 0.4, 0.2, 0.1   # data/transforms.py:33
JITTER_P, BLUR_P, FLIP_P = 0.5, 0.2, 0.5                     # data/transforms.py:31-36
BLUR_SIGMA = (0.1, 2.0)                                      # torchvision GaussianBlur default sigma range


def grayscale(img):
    """ITU-R 601-2 luma, as torchvision's rgb_to_grayscale / PIL mode 'L'.  img [3,H,W] -> [1,H,W]."""
    r, g, b = img[0], img[1], img[2]
    return (0.2989 * r + 0.587 * g + 0.114 * b)[None]


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0.0, 1.0)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    return _blend(img, grayscale(img).mean(), f)


def adjust_saturation(img, f):
    return _blend(img, grayscale(img), f)


def rgb_to_hsv(img):
    r, g, b = img[0], img[1], img[2]
    maxc = img.max(0).values
    minc = img.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc))


def hsv_to_rgb(img):
    h, s, v = img[0], img[1], img[2]
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = (v * (1.0 - s)).clamp(0.0, 1.0)
    q = (v * (1.0 - s * f)).clamp(0.0, 1.0)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0.0, 1.0)
    sel = [(v, q, p, p, t, v), (t, v, v, q, p, p), (p, p, t, v, v, q)]
    out = []
    for ch in sel:
        acc = torch.zeros_like(v)
        for k, c in enumerate(ch):
            acc = torch.where(i == k, c, acc)
        out.append(acc)
    return torch.stack(out)


def adjust_hue(img, f):
    hsv = rgb_to_hsv(img)
    h = (hsv[0] + f) % 1.0
    return hsv_to_rgb(torch.stack((h, hsv[1], hsv[2])))


_JITTER_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def color_jitter_params():
    """torchvision ColorJitter.get_params: op order (randperm(4)) and the four factors."""
    order = torch.randperm(4).tolist()
    b = float(torch.empty(1).uniform_(1.0 - BRIGHTNESS, 1.0 + BRIGHTNESS))
    c = float(torch.empty(1).uniform_(1.0 - CONTRAST, 1.0 + CONTRAST))
    s = float(torch.empty(1).uniform_(1.0 - SATURATION, 1.0 + SATURATION))
    h = float(torch.empty(1).uniform_(-HUE, HUE))
    return order, (b, c, s, h)


def color_jitter(img, order, factors):
    for k in order:
        img = _JITTER_OPS[k](img, factors[k])
    return img


def gaussian_kernel1d(sigma, ksize=3):
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize)
    pdf = torch.exp(-0.5 * (x / sigma) ** 2)
    return pdf / pdf.sum()


def gaussian_blur3(img, sigma):
    """3x3 separable Gaussian, reflect padding (torchvision functional_tensor.gaussian_blur)."""
    k1 = gaussian_kernel1d(sigma).to(img.device, img.dtype)
    k2 = (k1[:, None] * k1[None, :])[None, None].expand(3, 1, 3, 3)
    x = torch.nn.functional.pad(img[None], (1, 1, 1, 1), mode="reflect")
    return torch.nn.functional.conv2d(x, k2, groups=3)[0]


def draw_structure_params():
    """One draw of dino_structure_transforms' random decisions, in torchvision's call order:
    (flip, (order, factors) or None, sigma or None)."""
    flip = bool(torch.rand(1) < FLIP_P)
    jitter = None
    if not (JITTER_P < torch.rand(1)):
        jitter = color_jitter_params()
    sigma = None
    if not (BLUR_P < torch.rand(1)):
        sigma = float(torch.empty(1).uniform_(*BLUR_SIGMA))
    return flip, jitter, sigma


def apply_structure_torch(img, flip, jitter, sigma):
    """Reference implementation with torch ops (any device)."""
    if flip:
        img = img.flip(-1)
    if jitter is not None:
        img = color_jitter(img, *jitter)
    if sigma is not None:
        img = gaussian_blur3(img, sigma)
    return img


_scratch = {}


def apply_structure_hip(img, flip, jitter, sigma):
    """The same on the HIP kernels (splice_augment_structure): at most three launches for the whole pipeline."""
    import ctypes as C
    from . import _lib
    if not (flip or jitter is not None or sigma is not None):
        return img
    img = img.contiguous()
    _, H, W = img.shape
    key = (img.device, H, W)
    if key not in _scratch:
        _scratch[key] = torch.empty(3 * H * W + 256, device=img.device)
    out = torch.empty_like(img)
    order, factors = jitter if jitter is not None else ([], (1.0, 1.0, 1.0, 0.0))
    c_order = (C.c_int * 4)(*(list(order) + [0] * (4 - len(order))))
    c_fac = (C.c_float * 4)(*factors)
    _lib.check(_lib.lib().splice_augment_structure(_lib.ptr(img), _lib.ptr(out), _lib.ptr(_scratch[key]), H, W, int(flip), len(order),
                                                   c_order, c_fac, float(sigma) if sigma is not None else 0.0, _lib.current_stream()),
               "augment_structure")
    return out


def structure_transforms(img):
    """dino_structure_transforms on a [3,H,W] float tensor (whole image): HIP kernels on the GPU, torch ops otherwise."""
    flip, jitter, sigma = draw_structure_params()
    if img.is_cuda and img.dtype == torch.float32:
        return apply_structure_hip(img, flip, jitter, sigma)
    return apply_structure_torch(img, flip, jitter, sigma)


def texture_transforms(img):
    if torch.rand(1) < FLIP_P:
        img = img.flip(-1)
    return img


def global_crop_box(h, w, min_cover):
    """Global_crops.forward + RandomCrop.get_params: (top, left, size)."""
    size = int(round(np.random.uniform(min_cover * h, h)))   # data/transforms.py:21
    size = min(size, w)                                      # :22  (RandomCrop(int) is square)
    if size == h and size == w:
        return 0, 0, size
    top = int(torch.randint(0, h - size + 1, (1,)).item())
    left = int(torch.randint(0, w - size + 1, (1,)).item())
    return top, left, size


def global_crop_boxes(h, w, min_cover, n_crops):
    """``Global_crops.forward`` with ``n_crops`` crops (data/transforms.py:19-27): ONE size draw, then one RandomCrop position
    per crop.  Returns (size, [(top, left), ...])."""
    size = min(int(round(np.random.uniform(min_cover * h, h))), w)
    boxes = []
    for _ in range(n_crops):
        if size == h and size == w:
            boxes.append((0, 0))
        else:
            boxes.append((int(torch.randint(0, h - size + 1, (1,)).item()), int(torch.randint(0, w - size + 1, (1,)).item())))
    return size, boxes


def blur_sigma_to_weights(sigma):
    """(centre, side) weights of the 1-D kernel -- handy closed form for tests."""
    e = math.exp(-0.5 / (sigma * sigma))
    return 1.0 / (1.0 + 2.0 * e), e / (1.0 + 2.0 * e)
