#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own modules (build container only).

Run from the repo root:  ``python oracle/make_golden.py``  (needs /root/reference).

The reference cannot travel to the GPU box, so its outputs on seeded inputs are frozen
here as small data fixtures (inputs are regenerated from ``splice_amd.synth`` seeds, only
expected outputs / strided samples are stored).  What is driven:

  * ``models.extractor.attn_cosine_sim`` and ``VitExtractor.get_{queries,keys,values}_from_qkv``,
    ``get_keys_from_input``, ``get_keys_self_sim_from_input``, ``get_feature_from_input``
  * ``models.networks.define_G`` / ``models.unet.skip.skip`` forward + autograd backward (default architecture, and the
    6-scale reflection-padded net of ``inversion.py``)
  * ``util.losses.LossG`` + ``models.model.Model`` + ``util.util.get_optimizer`` for
    step losses, d loss / d params and a 20-step trajectory.

Two things the reference needs are absent from this image and are supplied as
stand-ins, which is why those parts stay "parity unpinned" (DESIGN.md):
  * ``torch.hub.load('facebookresearch/dino:main', ...)`` -> ``oracle.dino_vit``
    (our restatement; seeded synthetic weights),
  * ``torchvision.transforms`` {Resize, Normalize, Compose, ToPILImage} -> a tensor-only
    shim restating torchvision 0.10 semantics (``oracle.losses.resize_shorter_edge``).
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import dino_vit, losses as olosses  # noqa: E402
from oracle.fixtures import INVERSION_NET, sample, stats  # noqa: E402
from splice_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


# --------------------------------------------------------------------------- stand-ins
def install_torchvision_shim():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Resize:
        def __init__(self, size, max_size=None):
            self.size, self.max_size = size, max_size

        def __call__(self, img):
            return olosses.resize_shorter_edge(img, self.size, self.max_size)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, img):
            m = torch.tensor(self.mean, dtype=img.dtype).view(3, 1, 1)
            s = torch.tensor(self.std, dtype=img.dtype).view(3, 1, 1)
            return (img - m) / s

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, img):
            for t in self.ts:
                img = t(img)
            return img

    class ToPILImage:
        def __call__(self, img):
            raise RuntimeError("not used by the golden generator")

    tr.Resize, tr.Normalize, tr.Compose, tr.ToPILImage = Resize, Normalize, Compose, ToPILImage
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr


_VIT_CACHE = {}


def make_vit(model_name, img_size, seed=7, w_std=0.05):
    key = (model_name, img_size, seed, w_std)
    if key not in _VIT_CACHE:
        patch, dim, depth, heads = dino_vit.DINO_CONFIGS[model_name]
        m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=img_size).eval()
        sd = synth.vit_params(seed, model_name, img_size=img_size, w_std=w_std)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        _VIT_CACHE[key] = m
    return _VIT_CACHE[key]


HUB_IMG_SIZE = {"dino_vits8": 32}


def install_hub_stub():
    def fake_load(repo, model_name, *a, **k):
        assert repo == "facebookresearch/dino:main"
        return make_vit(model_name, HUB_IMG_SIZE[model_name])
    torch.hub.load = fake_load


# --------------------------------------------------------------------------- fixtures
def golden_extractor(ref_extractor_mod):
    out = {}
    for T, D in ((5, 8), (197, 64)):
        x = torch.from_numpy(synth.normal(11, f"cos/{T}", (1, 1, T, D)))
        out[f"cos_T{T}_D{D}"] = ref_extractor_mod.attn_cosine_sim(x).numpy()
    # zero row -> clamp path (product of norms below eps)
    x = torch.from_numpy(synth.normal(11, "cos/zero", (1, 1, 6, 8)).copy())
    x[0, 0, 2] = 0
    out["cos_zero_row"] = ref_extractor_mod.attn_cosine_sim(x).numpy()

    ext = ref_extractor_mod.VitExtractor("dino_vits8", "cpu")
    img_shape = (1, 3, 32, 32)
    qkv = torch.from_numpy(synth.normal(12, "qkv", (1, 17, 3 * 384)))
    out["q_from_qkv"] = ext.get_queries_from_qkv(qkv, img_shape).numpy()
    out["k_from_qkv"] = ext.get_keys_from_qkv(qkv, img_shape).numpy()
    out["v_from_qkv"] = ext.get_values_from_qkv(qkv, img_shape).numpy()

    img = torch.from_numpy(synth.normal(13, "img32", img_shape))
    feats = ext.get_feature_from_input(img)
    out["vits8_block_last"] = feats[-1].detach().numpy()
    out["vits8_block0"] = feats[0].detach().numpy()
    out["vits8_qkv11"] = ext.get_qkv_feature_from_input(img)[11].detach().numpy()
    out["vits8_attn11"] = ext.get_attn_feature_from_input(img)[11].detach().numpy()
    out["vits8_keys11"] = ext.get_keys_from_input(img, 11).detach().numpy()
    out["vits8_selfsim11"] = ext.get_keys_self_sim_from_input(img, 11).detach().numpy()
    # non-square input -> pos-embed interpolation path (K20): 32x48
    img2 = torch.from_numpy(synth.normal(13, "img32x48", (1, 3, 32, 48)))
    out["vits8_32x48_selfsim11"] = ext.get_keys_self_sim_from_input(img2, 11).detach().numpy()
    out["vits8_32x48_block_last"] = ext.get_feature_from_input(img2)[-1].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "extractor.npz"), **out)
    print("extractor.npz", {k: v.shape for k, v in out.items()})


def golden_generator(ref_networks_mod):
    out = {}
    params = synth.generator_params(21, 0.02, perturb_bias=0.05)
    net = ref_networks_mod.define_G("xavier", 0.02, initialize_weights=False)
    sd = net.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    net.load_state_dict(sd)
    names = [n for n, _ in net.named_parameters()]
    assert names == list(params.keys()), "parameter order differs from the reference"
    out["param_names"] = np.array(names)
    for tag, (h, w) in {"64x64": (64, 64), "213x213": (213, 213), "96x130": (96, 130)}.items():
        x = torch.from_numpy(synth.uniform(22, "gin/" + tag, (1, 3, h, w)))
        net.zero_grad()
        y = net(x)
        # a loss with a non-trivial gradient everywhere
        wgt = torch.from_numpy(synth.normal(23, "gw/" + tag, (1, 3, h, w)))
        loss = (y * wgt).sum() / y.numel() + (y * y).mean()
        loss.backward()
        out[f"{tag}/out_sample"] = sample(y, 4099)
        out[f"{tag}/out_stats"] = stats(y)
        out[f"{tag}/loss"] = np.float64(loss.item())
        if tag == "64x64":
            out[f"{tag}/out_full"] = y.detach().numpy()
        out[f"{tag}/grad_stats"] = np.stack([stats(p.grad) for p in net.parameters()])
        out[f"{tag}/grad_samples"] = np.stack([np.resize(sample(p.grad, 16), 16) for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, "generator.npz"), **out)
    print("generator.npz", len(out))


def golden_inversion_net():
    """The 6-scale reflection-padded skip() of inversion.py:21-25 (reference module), parameters seeded BY POSITION in
    parameters() order (conv weights N(0, 0.05), everything else 1 + N(0, 0.05) / N(0, 0.05)), forward + backward."""
    from models.unet.skip import skip as ref_skip
    net = ref_skip(8, 3, **INVERSION_NET)
    with torch.no_grad():
        for i, (name, p) in enumerate(net.named_parameters()):
            off = 1.0 if p.dim() == 1 and name.endswith("weight") else 0.0   # BatchNorm gains around 1
            p.copy_(torch.from_numpy(synth.normal(31, f"inv/p{i}", tuple(p.shape), 0.05, off)))
    out = {"n_params": np.int64(sum(p.numel() for p in net.parameters())), "n_tensors": np.int64(len(list(net.parameters())))}
    for tag, (h, w) in {"96x72": (96, 72), "100x84": (100, 84)}.items():   # the second one exercises the Concat centre crop
        x = torch.from_numpy(synth.normal(32, "inv/x" + tag, (1, 8, h, w)))
        net.zero_grad()
        y = net(x)
        (y * y).mean().backward()
        out[f"{tag}/out_sample"] = sample(y, 2053)
        out[f"{tag}/out_stats"] = stats(y)
        out[f"{tag}/grad_stats"] = np.stack([stats(p.grad) for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, "inversion_net.npz"), **out)
    print("inversion_net.npz", len(out))


def golden_define_g_init(ref_networks_mod):
    """Initial generator of ``torch.manual_seed(s); define_G(init_type, 0.02)`` (models/networks.py:24-58) for a few
    seeds / init types: per-tensor (sum, |sum|, sum sq) + a strided sample.  Pins the seed -> initial-weights map of
    splice_amd.networks (constructor draws + init draws on the CPU generator)."""
    out = {}
    for seed, init_type in ((0, "xavier"), (3, "xavier"), (5, "normal"), (7, "kaiming")):
        torch.manual_seed(seed)
        net = ref_networks_mod.define_G(init_type, 0.02)
        out[f"{init_type}/{seed}/stats"] = np.stack([stats(p) for p in net.parameters()])
        out[f"{init_type}/{seed}/samples"] = np.stack([np.resize(sample(p, 8), 8) for p in net.parameters()])
    np.savez_compressed(os.path.join(OUT, "define_g_init.npz"), **out)
    print("define_g_init.npz", len(out))


def golden_skip_constructor_init():
    """``torch.manual_seed(s); skip(...)`` as the reference leaves it WITHOUT init_weights (inversion.py:21-25 trains from this
    state): PyTorch's constructor initialisation of every nn.Conv2d / nn.BatchNorm2d, in the reference's construction order.
    Per-tensor (sum, |sum|, sum sq) + a strided sample, for the inversion net (32 noise channels) and the default arguments."""
    from models.unet.skip import skip as ref_skip
    out = {}
    for tag, seed, build in (("inversion", 2, lambda: ref_skip(32, 3, **INVERSION_NET)), ("inversion", 9, lambda: ref_skip(32, 3, **INVERSION_NET)),
                             ("default", 4, lambda: ref_skip())):
        torch.manual_seed(seed)
        net = build()
        ps = list(net.parameters())
        out[f"{tag}/{seed}/stats"] = np.stack([stats(p) for p in ps])
        out[f"{tag}/{seed}/samples"] = np.stack([np.resize(sample(p, 8), 8) for p in ps])
        out[f"{tag}/{seed}/next_draw"] = torch.rand(4).numpy()   # where the global generator stands afterwards
    np.savez_compressed(os.path.join(OUT, "skip_constructor_init.npz"), **out)
    print("skip_constructor_init.npz", len(out))


def run_reference_loop(cfg, A, B, n_steps, ref, A_entire=None, record_grads_at=()):
    """The body of ``train.py:51-80`` with the dataset replaced by fixed full crops
    (``Global_crops`` with min_cover=1 returns the whole image, data/transforms.py:22-23)."""
    Model, LossG, get_optimizer = ref
    model = Model(cfg)
    params = synth.generator_params(cfg["gen_seed"], cfg["init_gain"])
    sd = model.netG.state_dict()
    for k, v in params.items():
        sd[k] = torch.from_numpy(v)
    model.netG.load_state_dict(sd)
    criterion = LossG(cfg)
    for p in criterion.extractor.model.parameters():
        p.requires_grad_(False)
    optimizer = get_optimizer(cfg, model.netG.parameters())
    rec = {"losses": [], "grads": {}}
    step = -1
    for it in range(n_steps):
        step += 1
        inputs = {"step": step, "A_global": A[None], "B_global": B[None]}
        if step % cfg["entire_A_every"] == 0:
            inputs["A"] = (A if A_entire is None else A_entire)[None]
        optimizer.zero_grad()
        outputs = model(inputs)
        losses = criterion(outputs, inputs)
        losses["loss"].backward()
        row = {k: float(v) for k, v in losses.items()}
        rec["losses"].append(row)
        if step in record_grads_at:
            rec["grads"][step] = np.stack([stats(p.grad) for p in model.netG.parameters()])
        optimizer.step()
    with torch.no_grad():
        rec["final_out"] = model.netG(A[None]).numpy()
    return rec


LOSS_KEYS = ["loss", "loss_global_ssim", "loss_entire_ssim", "loss_entire_cls",
             "loss_global_cls", "loss_global_id_B"]


def pack_losses(rows):
    arr = np.full((len(rows), len(LOSS_KEYS)), np.nan, np.float64)
    for i, r in enumerate(rows):
        for j, k in enumerate(LOSS_KEYS):
            if k in r:
                arr[i, j] = r[k]
    return arr


def golden_steps(ref):
    base = dict(olosses.DEFAULT_CFG)
    out = {"loss_keys": np.array(LOSS_KEYS)}
    # (a) 64x64 pair (the smallest the 5-scale generator accepts: train-mode BN needs
    #     >1 value per channel at the 2x2 bottleneck), identity resize, vits8-shaped
    #     stand-in, T=65; 78 steps so that steps 0,1,2,75 (the four lambda regimes) are
    #     all covered.
    HUB_IMG_SIZE["dino_vits8"] = 64
    cfg = dict(base, dino_model_name="dino_vits8", dino_global_patch_size=64, gen_seed=31)
    A, B = synth.smooth_image_pair(32, 0, 64, 64)
    rec = run_reference_loop(cfg, torch.from_numpy(A), torch.from_numpy(B), 78, ref,
                             record_grads_at=(0, 1, 2, 75))
    out["a/losses"] = pack_losses(rec["losses"])
    out["a/final_out"] = rec["final_out"]
    for s, g in rec["grads"].items():
        out[f"a/grad_stats_step{s}"] = g
    # (b) 48x80 pair: non-identity resize (shorter edge 48 -> 64, long 80 -> 106) and
    #     non-square 8x13 token grid (pos-embed interpolation) ; 4 steps
    cfg = dict(base, dino_model_name="dino_vits8", dino_global_patch_size=64, gen_seed=33)
    A, B = synth.smooth_image_pair(34, 1, 48, 80)
    rec = run_reference_loop(cfg, torch.from_numpy(A), torch.from_numpy(B), 4, ref,
                             record_grads_at=(0, 1))
    out["b/losses"] = pack_losses(rec["losses"])
    out["b/final_out"] = rec["final_out"]
    for s, g in rec["grads"].items():
        out[f"b/grad_stats_step{s}"] = g
    np.savez_compressed(os.path.join(OUT, "steps.npz"), **out)
    print("steps.npz a-loss first/last", out["a/losses"][0, 0], out["a/losses"][-1, 0])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_torchvision_shim()
    install_hub_stub()
    sys.path.insert(0, REF)
    os.chdir(REF)
    import models.extractor as ref_extractor
    import models.networks as ref_networks
    from models.model import Model
    from util.losses import LossG
    from util.util import get_optimizer
    only = set(sys.argv[1:])   # e.g. `python oracle/make_golden.py inversion_net` regenerates one fixture
    if not only or "extractor" in only:
        golden_extractor(ref_extractor)
    if not only or "generator" in only:
        golden_generator(ref_networks)
    if not only or "steps" in only:
        golden_steps((Model, LossG, get_optimizer))
    if not only or "inversion_net" in only:
        golden_inversion_net()
    if not only or "define_g_init" in only:
        golden_define_g_init(ref_networks)
    if not only or "skip_constructor_init" in only:
        golden_skip_constructor_init()


if __name__ == "__main__":
    main()
