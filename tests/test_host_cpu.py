"""CPU: host-side logic of the product (no HIP calls): Resize output sizes, position-embedding
interpolation, the synthetic generators, the device data feed's crop law."""
import numpy as np
import pytest
import torch

from oracle import dino_vit, losses as olosses
from splice_amd import synth
from splice_amd.engine import resize_output_size
from splice_amd.vit import interpolate_pos_encoding


def test_resize_output_size_matches_oracle_resize():
    for h, w in [(224, 224), (213, 213), (128, 128), (48, 80), (80, 48), (224, 298), (900, 1200), (100, 400), (400, 100), (448, 448)]:
        for size in (224, 64, 448):
            out = olosses.resize_shorter_edge(torch.zeros(3, h, w), size, 480)
            assert tuple(out.shape[-2:]) == resize_output_size(h, w, size, 480), (h, w, size)


def test_pos_embed_interpolation_matches_oracle():
    m = dino_vit.VisionTransformer(8, 384, 1, 6, img_size=64).eval()
    with torch.no_grad():
        m.pos_embed.copy_(torch.from_numpy(synth.normal(3, "pos", tuple(m.pos_embed.shape))))
    for h, w in [(64, 64), (32, 48), (64, 106), (40, 40)]:
        x = torch.zeros(1, 1 + (h // 8) * (w // 8), 384)
        ref = m.interpolate_pos_encoding(x, h, w)[0]
        got = interpolate_pos_encoding(m.pos_embed.detach(), 8, h, w)
        assert got.shape == ref.shape
        assert torch.allclose(got, ref, atol=1e-6)


def test_synth_is_deterministic_and_distributed():
    a = synth.uniform(1, "x", (1000,))
    b = synth.uniform(1, "x", (1000,))
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert 0.0 <= a.min() and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.05
    assert not np.array_equal(a, synth.uniform(2, "x", (1000,)))
    assert not np.array_equal(a, synth.uniform(1, "y", (1000,)))
    n = synth.normal(1, "n", (20000,), std=2.0, mean=1.0)
    assert abs(n.mean() - 1.0) < 0.06 and abs(n.std() - 2.0) < 0.06
    A0, B0 = synth.image_pair(1234, 0, 16, 16)
    A1, _ = synth.image_pair(1234, 1, 16, 16)
    assert A0.shape == (3, 16, 16) and not np.array_equal(A0, B0) and not np.array_equal(A0, A1)


def test_generator_param_table_matches_reference_order(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "generator.npz"))
    names = [n for n, _, _ in synth.generator_param_specs()]
    assert names == [str(s) for s in g["param_names"]]
    assert sum(int(np.prod(s)) for _, s, _ in synth.generator_param_specs()) == 1037523


def test_device_data_feed_crop_law():
    """crop side ~ round(U(min_cover*h, h)) clipped to the width, square, inside the image
    (data/transforms.py:19-27); step counter and the every-75th 'A' entry (data/Dataset.py:62-70)."""
    from splice_amd import train
    cfg = dict(use_augmentations=True, entire_A_every=75, global_A_crops_min_cover=0.95, global_B_crops_min_cover=0.95,
               global_A_crops_n_crops=1, global_B_crops_n_crops=3)
    A = torch.rand(3, 100, 140)
    B = torch.rand(3, 120, 90)
    np.random.seed(0)
    torch.manual_seed(0)
    feed = train.DeviceDataFeed(cfg, A, B)
    sides_a, sides_b = [], []
    for i in range(200):
        s = feed.next()
        assert int(s['step']) == i
        assert ('A' in s) == (i % 75 == 0)
        a, b = s['A_global'], s['B_global']          # [n_crops, 3, side, side]: one side per call, one position per crop
        assert a.shape[:2] == (1, 3) and b.shape[:2] == (3, 3)
        assert a.shape[2] == a.shape[3] and b.shape[2] == b.shape[3]
        sides_a.append(a.shape[2])
        sides_b.append(b.shape[2])
    assert min(sides_a) >= 95 and max(sides_a) <= 100 and len(set(sides_a)) > 3
    assert set(sides_b) == {90}   # U(114,120) clipped to the width 90
    cfg['use_augmentations'] = False
    cfg['global_A_crops_min_cover'] = 1.0
    feed = train.DeviceDataFeed(cfg, A, A)
    s = feed.next()
    assert s['A_global'].shape == (1, 3, 100, 100)


# ---- device augmentations (splice_amd/augment.py) against PIL / colorsys: the arithmetic the reference's PIL pipeline
# (data/transforms.py:30-41) performs, up to its uint8 rounding (one or two grey levels per op)
def _pil(img):
    from PIL import Image
    return Image.fromarray((img.permute(1, 2, 0).numpy() * 255.0 + 0.5).astype(np.uint8))


def _unpil(im):
    return torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)


@pytest.mark.parametrize("factor", [0.6, 0.93, 1.4])
def test_color_jitter_ops_match_pil(factor):
    from PIL import ImageEnhance
    from splice_amd import augment
    img = _unpil(_pil(torch.from_numpy(synth.smooth_image_pair(3, 0, 40, 56)[0])))   # exactly representable in uint8
    tol = 2.5 / 255
    assert (augment.adjust_brightness(img, factor) - _unpil(ImageEnhance.Brightness(_pil(img)).enhance(factor))).abs().max() < tol
    assert (augment.adjust_contrast(img, factor) - _unpil(ImageEnhance.Contrast(_pil(img)).enhance(factor))).abs().max() < tol
    assert (augment.adjust_saturation(img, factor) - _unpil(ImageEnhance.Color(_pil(img)).enhance(factor))).abs().max() < tol


@pytest.mark.parametrize("shift", [-0.1, 0.037, 0.1])
def test_adjust_hue_matches_colorsys(shift):
    import colorsys
    from splice_amd import augment
    img = torch.from_numpy(synth.image_pair(9, 0, 12, 10)[0])
    out = augment.adjust_hue(img, shift)
    ref = np.empty((3, 12, 10), dtype=np.float64)
    a = img.numpy().astype(np.float64)
    for y in range(12):
        for x in range(10):
            h, s, v = colorsys.rgb_to_hsv(a[0, y, x], a[1, y, x], a[2, y, x])
            ref[:, y, x] = colorsys.hsv_to_rgb((h + shift) % 1.0, s, v)
    assert np.abs(out.numpy() - ref).max() < 2e-6
    # hue rotation keeps value (max channel) and saturation
    assert (out.max(0).values - img.max(0).values).abs().max() < 1e-6


def test_gaussian_blur3_is_separable_reflect():
    from splice_amd import augment
    img = torch.from_numpy(synth.image_pair(5, 0, 9, 11)[0])
    for sigma in (0.1, 0.8, 2.0):
        wc, ws = augment.blur_sigma_to_weights(sigma)
        assert abs(wc + 2 * ws - 1) < 1e-12
        pad = np.pad(img.numpy(), ((0, 0), (1, 1), (1, 1)), mode="reflect").astype(np.float64)
        rows = ws * pad[:, :, :-2] + wc * pad[:, :, 1:-1] + ws * pad[:, :, 2:]
        ref = ws * rows[:, :-2] + wc * rows[:, 1:-1] + ws * rows[:, 2:]
        assert np.abs(augment.gaussian_blur3(img, sigma).numpy() - ref).max() < 1e-6
    assert (augment.gaussian_blur3(img, 0.1) - img).abs().max() < 1e-6   # sigma 0.1: identity to 1e-22


def test_structure_pipeline_statistics():
    """Rates of the three random branches and ranges of their parameters (data/transforms.py:30-37)."""
    from splice_amd import augment
    torch.manual_seed(11)
    img = torch.from_numpy(synth.smooth_image_pair(3, 0, 24, 32)[0])
    n, flips, jit, blur = 2000, 0, 0, 0
    for _ in range(n):
        out = augment.structure_transforms(img)
        assert out.shape == img.shape and out.min() >= -1e-6 and out.max() <= 1 + 1e-6
        same, mirrored = torch.equal(out, img), torch.equal(out, img.flip(-1))
        if not (same or mirrored):
            jit += 1   # jitter and/or blur changed the pixels
    torch.manual_seed(12)
    for _ in range(n):
        flips += int(torch.equal(augment.texture_transforms(img), img.flip(-1)))
    assert abs(flips / n - 0.5) < 0.04
    assert abs(jit / n - (1 - 0.5 * 0.8)) < 0.04          # P(jitter or blur) = 1 - 0.5 * 0.8 = 0.6
    orders, facs = set(), []
    for _ in range(500):
        o, f = augment.color_jitter_params()
        orders.add(tuple(o)); facs.append(f)
    facs = np.array(facs)
    assert len(orders) == 24
    assert facs[:, 0].min() >= 0.6 and facs[:, 0].max() <= 1.4 and facs[:, 1].min() >= 0.6 and facs[:, 1].max() <= 1.4
    assert facs[:, 2].min() >= 0.8 and facs[:, 2].max() <= 1.2 and np.abs(facs[:, 3]).max() <= 0.1
    assert abs(facs[:, 0].mean() - 1.0) < 0.03 and abs(facs[:, 3].mean()) < 0.01


# ---- local DINO checkpoint loader (splice_amd/checkpoint.py), all four variants, both public file layouts
@pytest.mark.parametrize("name", ["dino_vits8", "dino_vits16", "dino_vitb8", "dino_vitb16"])
def test_checkpoint_loader_variants(name, tmp_path):
    from splice_amd import checkpoint
    patch, dim, depth, heads = synth.DINO_CONFIGS[name]
    specs = synth.vit_param_specs(patch, dim, depth, 224)
    flat = {n: torch.full(tuple(s), 0.5) for n, s, _ in specs}
    assert tuple(flat["pos_embed"].shape) == (1, 1 + (224 // patch) ** 2, dim)
    p1 = tmp_path / "backbone.pth"
    torch.save(flat, p1)
    got_name, sd = checkpoint.load_dino_checkpoint(str(p1))
    assert got_name == name and list(sd) == [n for n, _, _ in specs]
    assert all(v.dtype == np.float32 for v in sd.values())
    # *_full_checkpoint.pth layout: teacher / student with DDP + backbone prefixes and a projection head
    import argparse
    full = {"teacher": {"module.backbone." + k: v.half() for k, v in flat.items()}, "student": {}, "epoch": 100,
            "args": argparse.Namespace(arch="vit", patch_size=patch)}   # the public full checkpoints pickle their Namespace
    full["teacher"]["module.head.mlp.0.weight"] = torch.zeros(8, dim)
    full["teacher"]["module.head.last_layer.weight_g"] = torch.zeros(8, 1)
    p2 = tmp_path / "full.pth"
    torch.save(full, p2)
    got_name, sd2 = checkpoint.load_dino_checkpoint(str(p2), name)
    assert got_name == name and set(sd2) == set(sd) and sd2["cls_token"].dtype == np.float32
    other = "dino_vitb16" if name != "dino_vitb16" else "dino_vits8"
    with pytest.raises(ValueError, match="was requested"):
        checkpoint.load_dino_checkpoint(str(p1), other)
    bad = dict(flat)
    del bad["blocks.3.mlp.fc2.bias"]
    torch.save(bad, p1)
    with pytest.raises(ValueError, match="missing"):
        checkpoint.load_dino_checkpoint(str(p1))
    bad = dict(flat)
    bad["blocks.0.attn.qkv.weight"] = torch.zeros(3 * dim, dim + 1)
    torch.save(bad, p1)
    with pytest.raises(ValueError, match="shape"):
        checkpoint.load_dino_checkpoint(str(p1))


def test_async_result_writer(tmp_path):
    """util/util.py:55-59 semantics (ToPILImage: float [0,1] -> uint8 by truncation), written by the worker thread."""
    from PIL import Image
    from splice_amd.util import AsyncResultWriter, save_result
    w = AsyncResultWriter(str(tmp_path))
    imgs = [torch.from_numpy(synth.image_pair(3, i, 20, 28)[0]) for i in range(5)]
    for im in imgs:
        w.submit(im)
    w.close()
    got = np.asarray(Image.open(tmp_path / "out" / "output.png"))
    assert np.array_equal(got, (imgs[-1].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8))
    save_result(imgs[-1], str(tmp_path / "sync"))
    assert np.array_equal(got, np.asarray(Image.open(tmp_path / "sync" / "out" / "output.png")))
    assert not (tmp_path / "out" / "output.png.tmp").exists()


def test_async_result_writer_skips_intermediate_images_never_the_last(tmp_path):
    """Round 5: out/output.png is overwritten by every logged image, so intermediate ones that come faster than the writer's interval may be skipped
    (force=False); a forced one -- train_model's last logged image -- is always written, whatever came before it."""
    from PIL import Image
    from splice_amd.util import AsyncResultWriter
    w = AsyncResultWriter(str(tmp_path))
    assert w.min_interval > 0
    imgs = [torch.from_numpy(synth.image_pair(4, i, 20, 28)[0]) for i in range(6)]
    for im in imgs[:-1]:
        w.submit(im, force=False)          # back to back: only the first is inside the interval's budget
    w.submit(imgs[-1], force=True)
    w.close()
    assert w.skipped == 4
    got = np.asarray(Image.open(tmp_path / "out" / "output.png"))
    assert np.array_equal(got, (imgs[-1].numpy().transpose(1, 2, 0) * 255.0).astype(np.uint8))


def test_facade_factories_cpu():
    """util/util.py:8-39 conventions of the registries: every documented policy builds, unknown names are RETURNED as
    NotImplementedError (not raised), tensor2im casts arrays / passes foreign objects through."""
    from splice_amd.util import get_optimizer, get_scheduler, tensor2im
    w = [torch.nn.Parameter(torch.zeros(3))]
    cfg = dict(lr=2e-3, optimizer_beta1=0.0, optimizer_beta2=0.99)
    for name, cls in (("adam", torch.optim.Adam), ("rmsprop", torch.optim.RMSprop), ("sgd", torch.optim.SGD)):
        opt = get_optimizer(dict(cfg, optimizer=name), w)
        assert isinstance(opt, cls) and opt.param_groups[0]["lr"] == 2e-3
    assert get_optimizer(dict(cfg, optimizer="adam"), w).param_groups[0]["betas"] == (0.0, 0.99)
    assert isinstance(get_optimizer(dict(cfg, optimizer="lion"), w), NotImplementedError)
    opt = get_optimizer(dict(cfg, optimizer="sgd"), w)
    for policy in ("linear", "step", "plateau", "cosine", "none"):
        sch = get_scheduler(opt, policy, n_epochs=10, n_epochs_decay=4, lr_decay_iters=3)
        assert hasattr(sch, "step")
    lin = get_scheduler(get_optimizer(dict(cfg, optimizer="sgd"), w), "linear", n_epochs_decay=4)
    assert [round(lin.lr_lambdas[0](e), 3) for e in (0, 1, 5, 9)] == [1.0, 0.8, 0.0, 0]
    assert isinstance(get_scheduler(opt, "warmup"), NotImplementedError)
    img = torch.tensor([[[[0.5, 2.0]], [[-1.0, 0.25]], [[1.0, 0.0]]]])           # [1,3,1,2]
    assert tensor2im(img).tolist() == [[[127, 0, 255], [255, 63, 0]]]
    arr = np.array([[1.7, 2.2]])
    assert tensor2im(arr).dtype == np.uint8 and tensor2im("x") == "x"


def test_extractor_arch_table_matches_reference_substring_rules():
    """models/extractor.py:105-130 parses the model name on every call; the table + fallback must agree with those rules."""
    from splice_amd.extractor import _arch_of

    def rule(name):
        patch = 8 if "8" in name else 16
        small = ("s" in name) if "dino" in name else ("small" in name)
        return (patch, 6 if small else 12, 384 if small else 768)
    for name in ("dino_vits8", "dino_vits16", "dino_vitb8", "dino_vitb16", "vit_small_patch8_224", "vit_base_patch16_224", "dino_xcit_s8"):
        assert _arch_of(name) == rule(name), name


def test_round_weights_bf16_matches_the_engine_packing_rules():
    """oracle/dino_vit.py round_weights_bf16 (the "model the engine holds" of the free-run spot checks): every Linear / patch-embedding weight goes
    through bf16 once, the q rows of the QKV projection AFTER their scaling by d^-1/2 log2(e) (vit_engine.hip pack_qkv), everything else stays fp32."""
    import numpy as np
    import torch
    from oracle import dino_vit
    from splice_amd import synth
    st = synth.vit_params(3, "dino_vits8", img_size=32, w_std=0.05)
    dim = 384
    out = dino_vit.round_weights_bf16(st, dim)
    assert set(out) == set(st)
    c = (64 ** -0.5) * 1.4426950408889634
    for k, v in st.items():
        a, b = torch.from_numpy(np.asarray(v)), out[k]
        if k.endswith("attn.qkv.weight"):
            assert torch.equal(b[dim:], a[dim:].bfloat16().float())
            assert torch.equal((b[:dim] * c).bfloat16().float(), (a[:dim] * c).bfloat16().float())   # the scaled q rows are bf16 values
            assert not torch.equal(b[:dim], a[:dim].bfloat16().float())                               # ... which is NOT the rounding of the unscaled rows
        elif k.endswith(".weight") and a.dim() >= 2:
            assert torch.equal(b, a.bfloat16().float()), k
        else:
            assert torch.equal(b, a), k   # biases, LayerNorm parameters, class token, position embedding: untouched
