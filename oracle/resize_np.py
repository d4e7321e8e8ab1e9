"""TEST INFRASTRUCTURE (oracle): an INDEPENDENT restatement of the tensor ``Resize`` the reference applies inside its losses
(``util/losses.py:19-24``: ``transforms.Resize(dino_global_patch_size, max_size=480)`` from torchvision~=0.10, requirements.txt:2).

torchvision 0.10, ``functional_tensor.resize`` on a tensor image:
  * output size (``size`` an int): ``short, long = (w, h) if w <= h else (h, w)``; unchanged when ``short == size``;
    ``new_short, new_long = size, int(size * long / short)``; with ``max_size``: if ``new_long > max_size`` then
    ``new_short, new_long = int(max_size * new_short / new_long), max_size``;
  * pixels: ``torch.nn.functional.interpolate(img, size=[new_h, new_w], mode='bilinear', align_corners=False)`` -- antialias did
    not exist as an option for tensors in 0.10 (it is opt-in from 0.11 and on by default only from 0.17: NOT this behaviour).
  * ``align_corners=False`` bilinear (ATen ``area_pixel_compute_source_index``): ``src = (dst + 0.5) * (in / out) - 0.5``, negative
    values clamped to 0; ``i0 = floor(src)``, ``i1 = min(i0 + 1, in - 1)``, ``lambda = src - i0``; rows first or columns first is the
    same sum of four products.

Written with plain numpy loops over the output axes (no torch), so that ``oracle.losses.resize_shorter_edge`` (torch's kernel) and the
HIP kernels ``resize_bilinear_fwd/bwd`` are each pinned against something neither of them is: tests/test_oracle_golden.py and
tests/test_ops_gpu.py replay ``tests/golden/resize_np.npz`` (written by oracle/make_resize_golden.py).
"""
import numpy as np


def resize_output_size(h, w, size, max_size=480):
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_long, new_short) if w <= h else (new_short, new_long)


def _axis(n_in, n_out):
    """source index pair and weight of every output coordinate along one axis -- in float64: the DEFINITION.  ATen evaluates the same
    expression in float32, where a source coordinate near 900 resolves to 6e-5: implementations that agree with this file to ~1e-4 of the
    pixel range at 900-pixel planes (and to ~1e-6 at 128) are equally faithful; which float32 rounding (fused or not) they use is not pinned."""
    scale = float(n_in) / float(n_out)
    dst = np.arange(n_out, dtype=np.float64)
    src = np.maximum((dst + 0.5) * scale - 0.5, 0.0)
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    return i0, i1, src - i0


def bilinear_resize(img, out_h, out_w):
    """img [C, H, W] -> [C, out_h, out_w] float32, align_corners=False, no antialias."""
    img = np.asarray(img, dtype=np.float64)
    c, h, w = img.shape
    y0, y1, ly = _axis(h, out_h)
    x0, x1, lx = _axis(w, out_w)
    out = np.empty((c, out_h, out_w), np.float64)
    for oy in range(out_h):
        r0, r1 = img[:, y0[oy], :], img[:, y1[oy], :]
        top = r0[:, x0] * (1.0 - lx) + r0[:, x1] * lx
        bot = r1[:, x0] * (1.0 - lx) + r1[:, x1] * lx
        out[:, oy, :] = top * (1.0 - ly[oy]) + bot * ly[oy]
    return out.astype(np.float32)


def resize_shorter_edge(img, size, max_size=480):
    h, w = img.shape[-2:]
    oh, ow = resize_output_size(h, w, size, max_size)
    return np.asarray(img, np.float32) if (oh, ow) == (h, w) else bilinear_resize(img, oh, ow)
