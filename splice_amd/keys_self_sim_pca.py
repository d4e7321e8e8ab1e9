"""Counterpart of the reference's ``keys_self_sim_pca.py`` on the HIP extractor.

Visualises the structure descriptor of the paper: the self-similarity of the layer-``L`` keys of an image
(``VitExtractor.get_keys_self_sim_from_input``, ``models/extractor.py:158-163``) is reduced to 3 components with
PCA (``keys_self_sim_pca.py:31-34``), the CLS row dropped, the patch grid min-max normalised and up-sampled by the
patch size (``:37-45``).  Pre-processing = ``Resize(224)`` + ``ToTensor`` + ImageNet ``Normalize`` (``:14-24``).

    python -m splice_amd.keys_self_sim_pca --image_path limes.jpeg --save_path pca.png [--layer 11]
        [--dino_model_name dino_vitb8] [--checkpoint dino_vitbase8_pretrain.pth | --synthetic]

DINO weights come from a local ``.pth`` (``--checkpoint`` or ``SPLICE_DINO_CHECKPOINT``); ``--synthetic`` uses the
seeded synthetic weights (structure-free pictures, for smoke tests only).  PCA stays ``sklearn`` on the host, as in
the reference: it is a 3-component fit of a [T,T] matrix, not part of the hot path.
"""
from argparse import ArgumentParser

import numpy as np
import torch

from .extractor import VitExtractor
from .train import _load_image

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def keys_self_sim_pca_image(img, vit_extractor, layer=11):
    """img: float tensor ``[3,H,W]`` in [0,1] (shorter edge already 224).  Returns the uint8 ``[H',W',3]`` PCA picture
    with H' = (H // patch) * patch."""
    from sklearn.decomposition import PCA
    x = img[None].to(device)
    mean = torch.tensor(_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(_STD, device=device).view(1, 3, 1, 1)
    with torch.no_grad():
        keys_self_sim = vit_extractor.get_keys_self_sim_from_input((x - mean) / std, layer)
    ss = keys_self_sim[0].float().cpu().numpy()
    reduced = PCA(n_components=3).fit_transform(ss)
    p = vit_extractor.get_patch_size()
    gh, gw = vit_extractor.get_height_patch_num(x.shape), vit_extractor.get_width_patch_num(x.shape)
    grid = reduced[1:].reshape(gh, gw, 3)                       # row 0 is the [CLS] token
    grid = (grid - grid.min()) / max(grid.max() - grid.min(), 1e-12)
    from PIL import Image
    small = Image.fromarray(np.uint8(grid * 255))
    return np.asarray(small.resize((gw * p, gh * p), Image.BILINEAR))


def visualize(args):
    from PIL import Image
    img = _load_image(args.image_path, 224)
    ext = VitExtractor(args.dino_model_name, device, checkpoint=args.checkpoint, synthetic=args.synthetic)
    Image.fromarray(keys_self_sim_pca_image(img, ext, args.layer)).save(args.save_path)


if __name__ == '__main__':
    parser = ArgumentParser()
    parser.add_argument("--image_path", type=str, default='datasets/feature_visualization/limes.jpeg')
    parser.add_argument("--layer", type=int, default=11, help='Transformer layer from which to extract the feature, between 0-11')
    parser.add_argument("--dino_model_name", type=str, default='dino_vitb8', help='options: dino_vitb8 | dino_vits8 | dino_vitb16 | dino_vits16')
    parser.add_argument("--save_path", type=str, required=True)
    parser.add_argument("--checkpoint", type=str, default=None, help='local DINO .pth (default: $SPLICE_DINO_CHECKPOINT)')
    parser.add_argument("--synthetic", action="store_true", help='seeded synthetic weights (smoke tests)')
    visualize(parser.parse_args())
