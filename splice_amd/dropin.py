"""Make the reference's own import names resolve to the MI355X engine.

``import splice_amd.dropin`` (one line at the top of a notebook or script) registers, in ``sys.modules``,

    train                 -> splice_amd.train          (``from train import train_model``: Splice.ipynb cell 8)
    models.extractor      -> splice_amd.extractor      (train.py:4-7, inversion.py:1, keys_self_sim_pca.py:1)
    models.model          -> splice_amd.model
    models.networks       -> splice_amd.networks
    models.unet.skip      -> a module exposing ``skip``  (inversion.py:4)
    util.losses           -> splice_amd.losses
    util.util             -> splice_amd.util
    data.Dataset          -> a module exposing ``SingleImageDataset`` (train.py:4; device-side counterpart)
    data.transforms       -> a module exposing ``Global_crops`` and the two augmentation pipelines
    inversion, keys_self_sim_pca -> the splice_amd counterparts of the two side scripts

so that the literal import lines of the reference (``train.py:4-7``, ``Splice.ipynb`` cell 8, ``inversion.py:1-4``,
``keys_self_sim_pca.py:1``) run unchanged and bind the engine's objects.  Nothing of the reference is imported or needed.

Names that are already taken by foreign modules are NOT silently replaced: ``install()`` raises ``ImportError`` naming them
(a checkout of the reference on ``sys.path`` that was imported first), unless ``install(force=True)`` or
``SPLICE_DROPIN_FORCE=1`` in the environment.
``uninstall()`` removes exactly what ``install()`` added.  Importing this module calls ``install()``.
"""
import importlib
import os
import sys
import types

_ALIASES = {
    "train": "splice_amd.train",
    "models.extractor": "splice_amd.extractor",
    "models.model": "splice_amd.model",
    "models.networks": "splice_amd.networks",
    "util.losses": "splice_amd.losses",
    "util.util": "splice_amd.util",
    "inversion": "splice_amd.inversion",
    "keys_self_sim_pca": "splice_amd.keys_self_sim_pca",
}
_PACKAGES = ("models", "models.unet", "util", "data")
_MARK = "__splice_amd_dropin__"
_installed = []


def _package(name):
    m = types.ModuleType(name, f"splice_amd.dropin: stands in for the reference package `{name}`")
    m.__path__ = []          # a package: lets `import models.extractor` find the registered submodule
    setattr(m, _MARK, True)
    return m


def _skip_module():
    from . import networks
    m = types.ModuleType("models.unet.skip", "splice_amd.dropin: `skip(...)` of models/unet/skip.py:4-11 on the HIP generator engine")
    m.skip = networks.skip
    setattr(m, _MARK, True)
    return m


class SingleImageDataset:
    """``data.Dataset.SingleImageDataset(cfg)`` (data/Dataset.py:12-73) on the device data feed: ``dataset[0]`` advances the step
    counter and returns ``{'step', 'A_global', 'B_global'[, 'A']}`` as tensors (``train.py:53-55`` moves every value with ``.to``),
    ``get_A()`` the whole structure image ``[1,3,H,W]``, ``len(dataset) == 1``.  Images live on the GPU; the augmentation pipelines
    are the device counterparts (``splice_amd/augment.py``)."""

    def __init__(self, cfg):
        import torch
        from . import train
        self.cfg = cfg
        A = train._load_image(train._first_file(os.path.join(cfg['dataroot'], 'A')), cfg['A_resize'])
        B = train._load_image(train._first_file(os.path.join(cfg['dataroot'], 'B')), cfg['B_resize'])
        if cfg['direction'] == 'BtoA':
            A, B = B, A
        print("Image sizes %s and %s" % (str((A.shape[2], A.shape[1])), str((B.shape[2], B.shape[1]))))
        self._feed = train.DeviceDataFeed(cfg, A, B)
        self._torch = torch

    @property
    def step(self):
        return self._torch.zeros(1) + self._feed.step

    def get_A(self):
        return self._feed.get_A()

    def __getitem__(self, index):
        sample = self._feed.next()
        sample['step'] = self._torch.zeros(1) + sample['step']
        return sample

    def __len__(self):
        return 1


class Global_crops:
    """``data.transforms.Global_crops(n_crops, min_cover, last_transform, flip=False)`` on ``[3,H,W]`` tensors: one square size
    ``int(round(U(min_cover*h, h)))`` clipped to the width per call, one random position per crop (data/transforms.py:19-27)."""

    def __init__(self, n_crops, min_cover, last_transform=None, flip=False):
        self.n_crops, self.min_cover, self.last_transform, self.flip = n_crops, min_cover, last_transform, flip

    def __call__(self, img):
        import torch
        from . import augment
        _, h, w = img.shape
        size, boxes = augment.global_crop_boxes(h, w, self.min_cover, self.n_crops)
        crops = []
        for top, left in boxes:
            c = img[:, top:top + size, left:left + size]
            if self.last_transform is not None:
                c = self.last_transform(c)
            if self.flip and torch.rand(1).item() < 0.5:
                c = c.flip(-1)
            crops.append(c)
        return torch.stack(crops).contiguous()

    forward = __call__


def _data_modules():
    from . import augment
    ds = types.ModuleType("data.Dataset", "splice_amd.dropin: device-side SingleImageDataset")
    ds.SingleImageDataset = SingleImageDataset
    tr = types.ModuleType("data.transforms", "splice_amd.dropin: device-side crops and augmentation pipelines")
    tr.Global_crops = Global_crops
    tr.dino_structure_transforms = augment.structure_transforms
    tr.dino_texture_transforms = augment.texture_transforms
    ds.Global_crops, ds.dino_structure_transforms, ds.dino_texture_transforms = tr.Global_crops, tr.dino_structure_transforms, tr.dino_texture_transforms
    for m in (ds, tr):
        setattr(m, _MARK, True)
    return {"data.Dataset": ds, "data.transforms": tr}


def _ours(mod):
    return getattr(mod, _MARK, False) or getattr(mod, "__name__", "").startswith("splice_amd")


_built = {}        # alias -> module object: built ONCE, so that a second install() binds the same objects (module identity is stable)
_displaced = {}    # alias -> the foreign module install(force=True) replaced; uninstall() puts it back


def install(force=False):
    """Register the aliases.  Idempotent: the alias modules are built once and a name already bound to ours is left alone, so module
    identity does not change between calls.  Raises ImportError if a name is already bound to a module that is not ours; with
    ``force=True`` the foreign module is remembered and ``uninstall()`` restores it."""
    names = list(_PACKAGES) + list(_ALIASES) + ["models.unet.skip", "data.Dataset", "data.transforms"]
    taken = [n for n in names if n in sys.modules and not _ours(sys.modules[n])]
    if taken and not force:
        raise ImportError("splice_amd.dropin: these module names are already imported from elsewhere (a reference checkout on "
                          f"sys.path?): {taken}; import splice_amd.dropin first, or call splice_amd.dropin.install(force=True)")
    if not _built:
        _built.update({name: _package(name) for name in _PACKAGES})
        _built.update({alias: importlib.import_module(target) for alias, target in _ALIASES.items()})
        _built["models.unet.skip"] = _skip_module()
        _built.update(_data_modules())
    for name, mod in _built.items():
        cur = sys.modules.get(name)
        if cur is mod or (cur is not None and _ours(cur)):
            continue                      # already ours (this call is a repeat, or the package itself was imported under that name)
        if cur is not None:
            _displaced[name] = cur        # force=True: a foreign module gives way, and comes back at uninstall()
        sys.modules[name] = mod
        _installed.append(name)
    # attribute access through the parent (`import models.extractor; models.extractor.VitExtractor`)
    for name in _built:
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, sys.modules[name])
    return sorted(_built)


def uninstall():
    """Remove what install() bound; foreign modules displaced by ``install(force=True)`` are restored."""
    while _installed:
        name = _installed.pop()
        sys.modules.pop(name, None)
        if name in _displaced:
            sys.modules[name] = _displaced.pop(name)


install(force=os.environ.get("SPLICE_DROPIN_FORCE") == "1")
