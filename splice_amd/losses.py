"""Drop-in for ``util/losses.py``: ``LossG(cfg)`` with ``.extractor``, ``.global_transform``,
``.lambdas``, ``update_lambda_config``, ``forward(outputs, inputs) -> dict`` and the three
``calculate_*`` methods, composed from the HIP ops (ViT features, key self-similarity, bilinear
Resize with its adjoint).  This is the API-parity path (one ViT forward per reference call site,
autograd between ops); ``train_model`` uses the fused ``SpliceEngine`` step instead.
"""
import torch
import torch.nn.functional as F

from . import _lib
from .engine import resize_output_size
from .extractor import VitExtractor

device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, oh, ow):
        c, h, w = img.shape
        out = torch.empty(c, oh, ow, device=img.device)
        _lib.check(_lib.lib().splice_resize_bilinear_fwd(_lib.ptr(img), _lib.ptr(out), c, h, w, oh, ow, _lib.current_stream()), "resize_fwd")
        ctx.dims = (c, h, w, oh, ow)
        return out

    @staticmethod
    def backward(ctx, dout):
        c, h, w, oh, ow = ctx.dims
        din = torch.empty(c, h, w, device=dout.device)
        dout = dout.contiguous().float()
        _lib.check(_lib.lib().splice_resize_bilinear_bwd(_lib.ptr(dout), _lib.ptr(din), c, h, w, oh, ow, _lib.current_stream()), "resize_bwd")
        return din, None, None


class GlobalTransform:
    """``transforms.Compose([Resize(size, max_size=480), Normalize(imagenet)])`` of util/losses.py:19-24
    for ``[3,H,W]`` CUDA tensors (torchvision 0.10 tensor semantics: bilinear, no antialias)."""

    def __init__(self, size, max_size=480):
        self.size, self.max_size = size, max_size

    def __call__(self, img):
        c, h, w = img.shape
        oh, ow = resize_output_size(h, w, self.size, self.max_size)
        if (oh, ow) != (h, w):
            img = _Resize.apply(img.contiguous().float(), oh, ow)
        mean = torch.tensor(IMAGENET_MEAN, device=img.device).view(3, 1, 1)
        std = torch.tensor(IMAGENET_STD, device=img.device).view(3, 1, 1)
        return (img - mean) / std


# One row per entry of the dict LossG.forward returns (util/losses.py:46-72), in the reference's evaluation order:
# (dict key, lambda name, feature compared, key of the generated batch in `outputs`, key of the target batch in `inputs`)
_TERMS = (
    ("loss_global_ssim", "lambda_global_ssim", "self_sim", "x_global", "A_global"),
    ("loss_entire_ssim", "lambda_entire_ssim", "self_sim", "x_entire", "A"),
    ("loss_entire_cls", "lambda_entire_cls", "cls", "x_entire", "B_global"),
    ("loss_global_cls", "lambda_global_cls", "cls", "x_global", "B_global"),
    ("loss_global_id_B", "lambda_global_identity", "keys", "y_global", "B_global"),
)
_SCHEDULED_AT_WARMUP = ("lambda_global_ssim", "lambda_global_identity")
_SCHEDULED_PERIODIC = ("lambda_entire_ssim", "lambda_entire_cls")
_LAST_LAYER = 11


class LossG(torch.nn.Module):
    """``util/losses.py:11-105``.  The five terms are rows of ``_TERMS``; each compares ONE DINO feature of a generated
    image with the same feature of a target image (no gradient into the target) crop by crop, and the three public
    ``calculate_*`` methods are that comparison with the feature fixed."""

    def __init__(self, cfg, extractor=None, **extractor_kwargs):
        super().__init__()
        self.cfg = cfg
        self.extractor = extractor or VitExtractor(model_name=cfg['dino_model_name'], device=device, **extractor_kwargs)
        self.global_transform = GlobalTransform(cfg['dino_global_patch_size'], max_size=480)
        # only the [CLS] appearance term is live before the warm-up step (util/losses.py:25-32)
        self.lambdas = {name: 0 for _, name, _, _, _ in _TERMS}
        self.lambdas['lambda_global_cls'] = cfg['lambda_global_cls']
        self._features = {
            "self_sim": lambda img: self.extractor.get_keys_self_sim_from_input(img, layer_num=_LAST_LAYER),
            "cls": lambda img: self.extractor.get_feature_from_input(img)[-1][0, 0, :],
            "keys": lambda img: self.extractor.get_keys_from_input(img, _LAST_LAYER),
        }

    def update_lambda_config(self, step):
        """Lambda schedule of util/losses.py:34-44: structure + identity terms are switched on for good at
        ``cls_warmup``; the entire-image terms are live only on multiples of ``entire_A_every``."""
        cfg = self.cfg
        if step == cfg['cls_warmup']:
            self.lambdas.update((name, cfg[name]) for name in _SCHEDULED_AT_WARMUP)
        entire_step = step % cfg['entire_A_every'] == 0
        self.lambdas.update((name, cfg[name] if entire_step else 0) for name in _SCHEDULED_PERIODIC)

    def forward(self, outputs, inputs):
        self.update_lambda_config(int(inputs['step']))
        losses, total = {}, 0
        for key, lam_name, feature, out_key, in_key in _TERMS:
            lam = self.lambdas[lam_name]
            if lam > 0:
                losses[key] = self._feature_mse(feature, outputs[out_key], inputs[in_key])
                total = total + lam * losses[key]
        losses['loss'] = total
        return losses

    def _feature_mse(self, feature, generated, targets):
        """sum over crops of mse(feature(T(generated_i)), feature(T(target_i))), target side under no_grad."""
        f = self._features[feature]
        acc = 0.0
        for gen_img, tgt_img in zip(generated, targets):   # one crop at a time: the extractor is batch-1
            with torch.no_grad():
                want = f(self.global_transform(tgt_img).unsqueeze(0).to(device))
            got = f(self.global_transform(gen_img).unsqueeze(0).to(device))
            acc = acc + F.mse_loss(got, want)
        return acc

    # the reference's public per-term entry points (argument order as in util/losses.py:74,85,96)
    def calculate_global_ssim_loss(self, outputs, inputs):
        return self._feature_mse("self_sim", outputs, inputs)

    def calculate_crop_cls_loss(self, outputs, inputs):
        return self._feature_mse("cls", outputs, inputs)

    def calculate_global_id_loss(self, outputs, inputs):
        return self._feature_mse("keys", outputs, inputs)
