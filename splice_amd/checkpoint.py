"""Local DINO checkpoint loader.

The reference gets its ViT from ``torch.hub.load('facebookresearch/dino:main', model_name)``
(``models/extractor.py:20``), which downloads ``hubconf.py`` and a ``*_pretrain.pth`` -- impossible without a
network and undesirable in production.  This module turns a ``.pth`` already on disk into the flat
``name -> fp32 array`` mapping ``VitEngine.load_state_dict`` / ``splice_vit_set_param`` consume, for all four
variants the reference can name (``conf/default/config.yaml:25``; ``models/extractor.py:105-130`` parse the patch
size / head count / width out of the same strings):

    dino_vits8  dino_vits16  dino_vitb8  dino_vitb16

Accepted files
  * the backbone-only checkpoints the hub entry points load (``dino_deitsmall8_pretrain.pth``,
    ``dino_vitbase16_pretrain.pth`` ...): a flat state dict with the key names listed in SURVEY.md section 8c;
  * the ``*_full_checkpoint.pth`` training checkpoints: a dict holding ``teacher`` / ``student`` state dicts whose
    keys carry ``module.`` / ``backbone.`` prefixes and a projection ``head`` (the teacher is used, as the hub does);
  * anything ``torch.save``d from a ``VisionTransformer.state_dict()``.
``head.*`` entries (DINO's projection head, absent from the hub model: ``num_classes=0``) are dropped.  The position
table is kept at its trained grid; ``splice_amd.vit.interpolate_pos_encoding`` resamples it per input shape.
"""
import numpy as np
import torch

from .synth import DINO_CONFIGS, vit_param_specs

_PREFIXES = ("module.", "backbone.")


def _strip(name):
    changed = True
    while changed:
        changed = False
        for p in _PREFIXES:
            if name.startswith(p):
                name, changed = name[len(p):], True
    return name


def normalize_state_dict(obj):
    """Checkpoint object -> flat {name: tensor} with DINO's VisionTransformer key names."""
    if isinstance(obj, dict):
        for k in ("teacher", "state_dict", "model", "student"):
            if k in obj and isinstance(obj[k], dict):
                obj = obj[k]
                break
    if not isinstance(obj, dict):
        raise ValueError("DINO checkpoint: expected a state dict (or a dict holding 'teacher' / 'state_dict' / 'model')")
    out = {}
    for k, v in obj.items():
        k = _strip(k)
        if k.startswith("head"):
            continue
        if torch.is_tensor(v) or isinstance(v, np.ndarray):
            out[k] = v
    return out


def infer_model_name(sd):
    """Variant from the tensor shapes: width from cls_token, patch from the patch-embed kernel."""
    dim = int(sd["cls_token"].shape[-1])
    patch = int(sd["patch_embed.proj.weight"].shape[-1])
    for name, (p, d, _, _) in DINO_CONFIGS.items():
        if p == patch and d == dim:
            return name
    raise ValueError(f"DINO checkpoint: no known variant with patch {patch} and width {dim}")


def validate(sd, model_name):
    patch, dim, depth, _ = DINO_CONFIGS[model_name]
    n_pos = int(sd["pos_embed"].shape[-2]) if "pos_embed" in sd else 0
    side = int(round((n_pos - 1) ** 0.5))
    if n_pos < 2 or side * side != n_pos - 1:
        raise ValueError(f"DINO checkpoint: pos_embed holds {n_pos} positions, expected 1 + a square grid")
    want = {n: tuple(s) for n, s, _ in vit_param_specs(patch, dim, depth, side * patch)}
    missing = sorted(set(want) - set(sd))
    extra = sorted(set(sd) - set(want))
    if missing:
        raise ValueError(f"DINO checkpoint is not a {model_name}: missing {missing[:4]}{' ...' if len(missing) > 4 else ''}")
    if extra:
        raise ValueError(f"DINO checkpoint is not a {model_name}: unexpected entries {extra[:4]}{' ...' if len(extra) > 4 else ''}")
    for n, shp in want.items():
        if tuple(sd[n].shape) != shp:
            raise ValueError(f"DINO checkpoint is not a {model_name}: {n} has shape {tuple(sd[n].shape)}, expected {shp}")


def _load_pickle(path):
    """``torch.load`` with ``weights_only`` stated explicitly.  Backbone-only files (``dino_*_pretrain.pth``) are plain
    tensor dicts.  The ``*_full_checkpoint.pth`` files additionally pickle the training ``argparse.Namespace``: those are
    retried with Namespace allow-listed (still no arbitrary code execution)."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as first:
        import argparse
        try:
            with torch.serialization.safe_globals([argparse.Namespace]):
                return torch.load(path, map_location="cpu", weights_only=True)
        except Exception:
            raise ValueError(f"{path}: cannot be read as a tensor-only checkpoint ({type(first).__name__}: {first}); "
                             "re-save it as a plain state dict (torch.save(model.state_dict(), ...))") from first


def load_dino_checkpoint(path, model_name=None):
    """Read ``path`` and return ``(model_name, state)`` with ``state`` = ordered {name: fp32 numpy array} ready for
    ``VitEngine.load_state_dict``.  ``model_name`` (one of DINO_CONFIGS) is checked against the file when given and
    inferred from the shapes otherwise."""
    obj = _load_pickle(path)
    sd = normalize_state_dict(obj)
    if "cls_token" not in sd or "patch_embed.proj.weight" not in sd:
        raise ValueError(f"{path}: not a DINO VisionTransformer checkpoint (no cls_token / patch_embed.proj.weight)")
    found = infer_model_name(sd)
    if model_name is not None and model_name != found:
        raise ValueError(f"{path}: holds a {found}, but {model_name} was requested (conf/default/config.yaml dino_model_name)")
    validate(sd, found)
    state = {k: torch.as_tensor(v).detach().to(torch.float32).contiguous().numpy() for k, v in sd.items()}
    return found, state
