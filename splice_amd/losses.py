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


class LossG(torch.nn.Module):
    def __init__(self, cfg, extractor=None, **extractor_kwargs):
        super().__init__()
        self.cfg = cfg
        self.extractor = extractor or VitExtractor(model_name=cfg['dino_model_name'], device=device, **extractor_kwargs)
        self.global_transform = GlobalTransform(cfg['dino_global_patch_size'], max_size=480)
        self.lambdas = dict(
            lambda_global_cls=cfg['lambda_global_cls'],
            lambda_global_ssim=0,
            lambda_entire_ssim=0,
            lambda_entire_cls=0,
            lambda_global_identity=0
        )

    def update_lambda_config(self, step):
        if step == self.cfg['cls_warmup']:
            self.lambdas['lambda_global_ssim'] = self.cfg['lambda_global_ssim']
            self.lambdas['lambda_global_identity'] = self.cfg['lambda_global_identity']
        if step % self.cfg['entire_A_every'] == 0:
            self.lambdas['lambda_entire_ssim'] = self.cfg['lambda_entire_ssim']
            self.lambdas['lambda_entire_cls'] = self.cfg['lambda_entire_cls']
        else:
            self.lambdas['lambda_entire_ssim'] = 0
            self.lambdas['lambda_entire_cls'] = 0

    def forward(self, outputs, inputs):
        self.update_lambda_config(int(inputs['step']))
        losses = {}
        loss_G = 0
        if self.lambdas['lambda_global_ssim'] > 0:
            losses['loss_global_ssim'] = self.calculate_global_ssim_loss(outputs['x_global'], inputs['A_global'])
            loss_G += losses['loss_global_ssim'] * self.lambdas['lambda_global_ssim']
        if self.lambdas['lambda_entire_ssim'] > 0:
            losses['loss_entire_ssim'] = self.calculate_global_ssim_loss(outputs['x_entire'], inputs['A'])
            loss_G += losses['loss_entire_ssim'] * self.lambdas['lambda_entire_ssim']
        if self.lambdas['lambda_entire_cls'] > 0:
            losses['loss_entire_cls'] = self.calculate_crop_cls_loss(outputs['x_entire'], inputs['B_global'])
            loss_G += losses['loss_entire_cls'] * self.lambdas['lambda_entire_cls']
        if self.lambdas['lambda_global_cls'] > 0:
            losses['loss_global_cls'] = self.calculate_crop_cls_loss(outputs['x_global'], inputs['B_global'])
            loss_G += losses['loss_global_cls'] * self.lambdas['lambda_global_cls']
        if self.lambdas['lambda_global_identity'] > 0:
            losses['loss_global_id_B'] = self.calculate_global_id_loss(outputs['y_global'], inputs['B_global'])
            loss_G += losses['loss_global_id_B'] * self.lambdas['lambda_global_identity']
        losses['loss'] = loss_G
        return losses

    def calculate_global_ssim_loss(self, outputs, inputs):
        loss = 0.0
        for a, b in zip(inputs, outputs):  # one crop at a time, as the reference
            a = self.global_transform(a)
            b = self.global_transform(b)
            with torch.no_grad():
                target_keys_self_sim = self.extractor.get_keys_self_sim_from_input(a.unsqueeze(0), layer_num=11)
            keys_ssim = self.extractor.get_keys_self_sim_from_input(b.unsqueeze(0), layer_num=11)
            loss += F.mse_loss(keys_ssim, target_keys_self_sim)
        return loss

    def calculate_crop_cls_loss(self, outputs, inputs):
        loss = 0.0
        for a, b in zip(outputs, inputs):
            a = self.global_transform(a).unsqueeze(0).to(device)
            b = self.global_transform(b).unsqueeze(0).to(device)
            cls_token = self.extractor.get_feature_from_input(a)[-1][0, 0, :]
            with torch.no_grad():
                target_cls_token = self.extractor.get_feature_from_input(b)[-1][0, 0, :]
            loss += F.mse_loss(cls_token, target_cls_token)
        return loss

    def calculate_global_id_loss(self, outputs, inputs):
        loss = 0.0
        for a, b in zip(inputs, outputs):
            a = self.global_transform(a)
            b = self.global_transform(b)
            with torch.no_grad():
                keys_a = self.extractor.get_keys_from_input(a.unsqueeze(0), 11)
            keys_b = self.extractor.get_keys_from_input(b.unsqueeze(0), 11)
            loss += F.mse_loss(keys_a, keys_b)
        return loss
