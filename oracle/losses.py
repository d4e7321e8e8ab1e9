"""Restatement of ``util/losses.py`` (test infrastructure only)."""
import torch
import torch.nn.functional as F

from . import extractor as ex

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

DEFAULT_CFG = dict(  # conf/default/config.yaml
    init_type="xavier", init_gain=0.02,
    lambda_global_cls=10.0, lambda_global_ssim=1.0, lambda_global_identity=1.0,
    entire_A_every=75, lambda_entire_cls=10, lambda_entire_ssim=1.0,
    dino_model_name="dino_vitb8", dino_global_patch_size=224,
    cls_warmup=1, n_epochs=10000, scheduler_policy="none",
    optimizer="adam", optimizer_beta1=0.0, optimizer_beta2=0.99, lr=0.002,
    log_images_freq=10)


def resize_shorter_edge(img, size, max_size=480):
    """torchvision 0.10 ``Resize(size, max_size)`` on a ``[C,H,W]`` tensor
    (``util/losses.py:20``): shorter edge -> ``size`` keeping aspect (long edge
    ``int(size*long/short)``, capped at ``max_size`` by shrinking both), bilinear,
    align_corners=False, NO antialias; identity when the shorter edge already matches."""
    h, w = img.shape[-2:]
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return img
    new_short, new_long = size, int(size * long / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    return F.interpolate(img[None], size=(nh, nw), mode="bilinear", align_corners=False)[0]


def normalize(img):
    mean = torch.tensor(IMAGENET_MEAN, dtype=img.dtype).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=img.dtype).view(3, 1, 1)
    return (img - mean) / std


def global_transform(img, cfg):
    """``util/losses.py:22-24``: Resize then Normalize, inside the autograd graph."""
    return normalize(resize_shorter_edge(img, cfg["dino_global_patch_size"], 480))


def initial_lambdas(cfg):
    """``util/losses.py:26-32``."""
    return dict(lambda_global_cls=cfg["lambda_global_cls"], lambda_global_ssim=0,
                lambda_entire_ssim=0, lambda_entire_cls=0, lambda_global_identity=0)


def update_lambdas(lambdas, cfg, step):
    """``util/losses.py:34-44`` (mutates and returns ``lambdas``)."""
    if step == cfg["cls_warmup"]:
        lambdas["lambda_global_ssim"] = cfg["lambda_global_ssim"]
        lambdas["lambda_global_identity"] = cfg["lambda_global_identity"]
    if step % cfg["entire_A_every"] == 0:
        lambdas["lambda_entire_ssim"] = cfg["lambda_entire_ssim"]
        lambdas["lambda_entire_cls"] = cfg["lambda_entire_cls"]
    else:
        lambdas["lambda_entire_ssim"] = 0
        lambdas["lambda_entire_cls"] = 0
    return lambdas


def ssim_loss(vit, cfg, outputs, inputs):
    """``util/losses.py:74-83``."""
    loss = 0.0
    for a, b in zip(inputs, outputs):
        a = global_transform(a, cfg)
        b = global_transform(b, cfg)
        with torch.no_grad():
            target = ex.keys_self_sim_from_input(vit, a.unsqueeze(0))
        loss = loss + F.mse_loss(ex.keys_self_sim_from_input(vit, b.unsqueeze(0)), target)
    return loss


def cls_loss(vit, cfg, outputs, inputs):
    """``util/losses.py:85-94``."""
    loss = 0.0
    for a, b in zip(outputs, inputs):
        a = global_transform(a, cfg).unsqueeze(0)
        b = global_transform(b, cfg).unsqueeze(0)
        cls = ex.cls_from_input(vit, a)
        with torch.no_grad():
            target = ex.cls_from_input(vit, b)
        loss = loss + F.mse_loss(cls, target)
    return loss


def id_loss(vit, cfg, outputs, inputs):
    """``util/losses.py:96-105``."""
    loss = 0.0
    for a, b in zip(inputs, outputs):
        a = global_transform(a, cfg)
        b = global_transform(b, cfg)
        with torch.no_grad():
            ka = ex.keys_from_input(vit, a.unsqueeze(0))
        kb = ex.keys_from_input(vit, b.unsqueeze(0))
        loss = loss + F.mse_loss(ka, kb)
    return loss


def loss_g(vit, cfg, lambdas, outputs, inputs):
    """``util/losses.py:46-72``.  ``inputs['step']`` is an int here."""
    update_lambdas(lambdas, cfg, int(inputs["step"]))
    losses, total = {}, 0
    if lambdas["lambda_global_ssim"] > 0:
        losses["loss_global_ssim"] = ssim_loss(vit, cfg, outputs["x_global"], inputs["A_global"])
        total = total + losses["loss_global_ssim"] * lambdas["lambda_global_ssim"]
    if lambdas["lambda_entire_ssim"] > 0:
        losses["loss_entire_ssim"] = ssim_loss(vit, cfg, outputs["x_entire"], inputs["A"])
        total = total + losses["loss_entire_ssim"] * lambdas["lambda_entire_ssim"]
    if lambdas["lambda_entire_cls"] > 0:
        losses["loss_entire_cls"] = cls_loss(vit, cfg, outputs["x_entire"], inputs["B_global"])
        total = total + losses["loss_entire_cls"] * lambdas["lambda_entire_cls"]
    if lambdas["lambda_global_cls"] > 0:
        losses["loss_global_cls"] = cls_loss(vit, cfg, outputs["x_global"], inputs["B_global"])
        total = total + losses["loss_global_cls"] * lambdas["lambda_global_cls"]
    if lambdas["lambda_global_identity"] > 0:
        losses["loss_global_id_B"] = id_loss(vit, cfg, outputs["y_global"], inputs["B_global"])
        total = total + losses["loss_global_id_B"] * lambdas["lambda_global_identity"]
    losses["loss"] = total
    return losses
