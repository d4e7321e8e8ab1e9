"""Counterpart of the reference's ``inversion.py`` on the HIP extractor.

Feature inversion (paper section 3 / ``inversion.py:12-74``): optimise a 6-scale skip network fed with fixed noise so
that the DINO feature of its output -- the layer-``L`` [CLS] token or the layer-``L`` keys -- matches the feature of a
reference image; for the [CLS] inversion the input noise is perturbed every iteration with a decaying amplitude
(x10 -> x2 -> x0.5, ``inversion.py:55-62``).  Every iteration is one ViT forward + backward through
``VitExtractor`` (the HIP engine, ``torch.autograd.Function`` around ``splice_vit_forward/backward``); the small
generator of this experiment is a stock-PyTorch ``skip(...)`` (``splice_amd/unet_general.py``).

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

    def extract_feature(x):   # the image is already 224 on its short edge: Resize(224) of inversion.py:29 is the identity
        x = (x - mean) / std
        if args.feature == 'cls':
            return ext.get_feature_from_input(x)[args.layer][:, 0, :]
        if args.feature == 'keys':
            return ext.get_keys_from_input(x, args.layer)
        raise ValueError('feature {} not supported.'.format(args.feature))

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


if __name__ == '__main__':
    parser = ArgumentParser()
    parser.add_argument("--feature", type=str, help='DINO-ViT feature to invert. options: cls | keys')
    parser.add_argument("--layer", type=int, default=11, help='Transformer layer from which to extract the feature, between 0-11')
    parser.add_argument("--dino_model_name", type=str, default='dino_vitb8')
    parser.add_argument("--image_path", type=str, default='datasets/feature_visualization/limes.jpeg', help='path to the image to be used for the inversion.')
    parser.add_argument("--save_path", type=str, required=True, help='path to save the result.')
    parser.add_argument("--log_freq", type=int, default=100)
    parser.add_argument("--input_depth", type=int, default=32)
    parser.add_argument("--LR", type=float, default=0.01)
    parser.add_argument("--n_iter", type=int, default=20000)
    parser.add_argument("--reduce_noise_stage_1_iter", type=int, default=10000)
    parser.add_argument("--reduce_noise_stage_2_iter", type=int, default=15000)
    parser.add_argument("--checkpoint", type=str, default=None, help='local DINO .pth (default: $SPLICE_DINO_CHECKPOINT)')
    parser.add_argument("--synthetic", action="store_true", help='seeded synthetic weights (smoke tests)')
    invert(parser.parse_args())
