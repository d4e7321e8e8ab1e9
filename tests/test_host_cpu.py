"""CPU: host-side logic of the product (no HIP calls): Resize output sizes, position-embedding
interpolation, the synthetic generators, the device data feed's crop law."""
import numpy as np
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
    cfg = dict(use_augmentations=True, entire_A_every=75, global_A_crops_min_cover=0.95, global_B_crops_min_cover=0.95)
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
        a, b = s['A_global'], s['B_global']
        assert a.shape[1] == a.shape[2] and b.shape[1] == b.shape[2]
        sides_a.append(a.shape[1])
        sides_b.append(b.shape[1])
    assert min(sides_a) >= 95 and max(sides_a) <= 100 and len(set(sides_a)) > 3
    assert set(sides_b) == {90}   # U(114,120) clipped to the width 90
    cfg['use_augmentations'] = False
    cfg['global_A_crops_min_cover'] = 1.0
    feed = train.DeviceDataFeed(cfg, A, A)
    s = feed.next()
    assert torch.equal(s['A_global'], A[:, :, :100]) or s['A_global'].shape == (3, 100, 100)
