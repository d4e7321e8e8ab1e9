"""Counterpart of the reference's ``inversion.py`` on the HIP extractor.

Feature inversion (paper section 3 / ``inversion.py:12-74``): optimise a 6-scale skip network fed with fixed noise so
that the DINO feature of its output -- the layer-``L`` [CLS] token or the layer-``L`` keys -- matches the feature of a
reference image; for the [CLS] inversion the input noise is perturbed every iteration with a decaying amplitude
(x10 -> x2 -> x0.5, ``inversion.py:55-62``).  Every iteration is one ViT forward + backward through
``VitExtractor`` (the HIP engine, ``torch.autograd.Function`` around ``splice_vit_forward/backward``) and one forward +
backward of the experiment's own generator -- ``skip()`` with 6 scales, 7/7/5/5/3/3 filters, reflection padding and noise
input -- on the HIP generator engine (``splice_gen_create_arch``; pinned against the reference's module in
tests/test_generator_gpu.py::test_inversion_net_on_hip_matches_reference_golden).

    python -m splice_amd.inversion --feature cls|keys --image_path img.jpg --save_path out.png
        [--layer 11] [--dino_model_name dino_vitb8] [--n_iter 20000] [--checkpoint dino.pth | --synthetic]
"""
from argparse import ArgumentParser

import numpy as np
import torch

from .extractor import VitExtractor
from .networks import skip
from .train import _load_image

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)

# feature name -> reader(extractor, normalised image [1,3,H,W], layer)
_FEATURE_READERS = {
    "cls": lambda ext, img, layer: ext.get_feature_from_input(img)[layer][:, 0, :],    # [CLS] row of the block output
    "keys": lambda ext, img, layer: ext.get_keys_from_input(img, layer),               # [heads, T, 64]
}
_REQUIRED = object()
_CLI = (  # (flag, type, default, help)
    ("feature", str, None, "which feature to invert: " + " | ".join(sorted(_FEATURE_READERS))),
    ("layer", int, 11, "index of the transformer block the feature is read from (0..11)"),
    ("dino_model_name", str, "dino_vitb8", "dino_vit{s,b}{8,16}"),
    ("image_path", str, "datasets/feature_visualization/limes.jpeg", "image whose feature is inverted"),
    ("save_path", str, _REQUIRED, "where the reconstructed image is written"),
    ("log_freq", int, 100, "write the current reconstruction every this many iterations"),
    ("input_depth", int, 32, "channels of the fixed noise the generator is fed with"),
    ("LR", float, 0.01, "Adam learning rate"),
    ("n_iter", int, 20000, "optimisation steps"),
    ("reduce_noise_stage_1_iter", int, 10000, "[cls] input-noise amplitude 10 -> 2 at this iteration"),
    ("reduce_noise_stage_2_iter", int, 15000, "[cls] input-noise amplitude 2 -> 0.5 at this iteration"),
    ("checkpoint", str, None, "local DINO .pth (default: $SPLICE_DINO_CHECKPOINT)"),
)


def make_net(input_depth):
    """``inversion.py:21-25``."""
    return skip(input_depth, 3, num_channels_down=[16, 32, 64, 128, 128, 128], num_channels_up=[16, 32, 64, 128, 128, 128],
                num_channels_skip=[4, 4, 4, 4, 4, 4], filter_size_down=[7, 7, 5, 5, 3, 3], filter_size_up=[7, 7, 5, 5, 3, 3],
                downsample_mode='stride', pad='reflection', device=device)


def noise_scale(i, stage1, stage2):
    """Amplitude of the per-iteration input perturbation of the [CLS] inversion (``inversion.py:55-62``)."""
    return 10.0 if i < stage1 else (2.0 if i < stage2 else 0.5)


def invert(args, callback=None):
    from PIL import Image
    input_img = _load_image(args.image_path, 224)[None].to(device)
    net = make_net(args.input_depth)
    net_input_saved = torch.randn((1, args.input_depth, input_img.shape[-2], input_img.shape[-1]), device=device)
    mean = torch.tensor(_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(_STD, device=device).view(1, 3, 1, 1)
    ext = VitExtractor(args.dino_model_name, device, checkpoint=getattr(args, "checkpoint", None), synthetic=getattr(args, "synthetic", False))

    if args.feature not in _FEATURE_READERS:
        raise ValueError(f"unknown --feature {args.feature!r}: choose one of {sorted(_FEATURE_READERS)}")
    read = _FEATURE_READERS[args.feature]

    def extract_feature(x):   # the image is already 224 on its short edge: Resize(224) of inversion.py:29 is the identity
        return read(ext, (x - mean) / std, args.layer)

    with torch.no_grad():
        ref_feature = extract_feature(input_img)
    optimizer = torch.optim.Adam(net.parameters(), lr=args.LR)
    losses = []
    for i in range(args.n_iter):
        net_input = net_input_saved
        if args.feature == 'cls':   # noise on the input as a regulariser, reduced in two stages
            net_input = net_input_saved + torch.randn_like(net_input_saved) * noise_scale(i, args.reduce_noise_stage_1_iter, args.reduce_noise_stage_2_iter)
        optimizer.zero_grad()
        loss = torch.nn.functional.mse_loss(extract_feature(net(net_input)), ref_feature)
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
        if i % args.log_freq == 0:
            with torch.no_grad():
                result = net(net_input)[0].clamp(0, 1)
            arr = (result.permute(1, 2, 0).cpu().numpy() * 255.0).astype(np.uint8)
            Image.fromarray(arr).save(args.save_path)
            if callback is not None:
                callback(i, losses[-1])
    return losses


def build_parser():
    """Same flags and defaults as the reference script (``inversion.py:78-92``) plus the two weight-source flags."""
    parser = ArgumentParser(description="Invert a DINO-ViT feature back to an image on the MI355X engine.")
    for flag, kind, default, doc in _CLI:
        kw = dict(type=kind, help=doc)
        if default is _REQUIRED:
            kw["required"] = True
        else:
            kw["default"] = default
        parser.add_argument("--" + flag, **kw)
    parser.add_argument("--synthetic", action="store_true", help="use the seeded synthetic ViT weights (smoke tests only)")
    return parser


if __name__ == '__main__':
    invert(build_parser().parse_args())
