"""GPU: splice_amd.batch on the real engine -- a directory of pairs through the worker queue (N = 1 worker process on the
one GPU of the test box) writes bit-identical images / losses to running train_model on each pair in THIS process."""
import os

import numpy as np
import pytest

from splice_amd import synth

pytestmark = pytest.mark.gpu
OVER = dict(seed=3, n_epochs=12, dino_model_name="dino_vits8", dino_global_patch_size=64, log_images_freq=6)


def _write_pairs(root, k, h=64, w=80):
    from PIL import Image
    for i in range(k):
        A, B = synth.smooth_image_pair(60, i, h, w)
        for side, img in (("A", A), ("B", B)):
            d = root / f"p{i}" / side
            d.mkdir(parents=True)
            Image.fromarray((img.transpose(1, 2, 0) * 255).astype(np.uint8)).save(d / "img.png")


def test_batch_queue_equals_serial_runs(tmp_path, monkeypatch):
    from splice_amd import batch
    from splice_amd.train import train_model
    monkeypatch.setenv("SPLICE_SYNTHETIC_WEIGHTS", "1")
    q, ser = tmp_path / "queue", tmp_path / "serial"
    for r in (q, ser):
        r.mkdir()
        _write_pairs(r, 2)
    res = batch.run_batch(str(q), 1, OVER)
    assert [r["pair"] for r in res] == ["p0", "p1"] and all(r["steps"] == OVER["n_epochs"] and r["gpu"] == 0 for r in res)
    for i, r in enumerate(res):
        eng = train_model(str(ser / f"p{i}"), cfg_overrides=OVER, progress=False)
        assert eng.losses()["loss"] == r["loss"]                     # same kernels, same order: bit-identical
        assert (ser / f"p{i}" / "out" / "output.png").read_bytes() == (q / f"p{i}" / "out" / "output.png").read_bytes()
    assert res[0]["loss"] != res[1]["loss"]


def test_pairs_in_one_step_equal_single_runs(tmp_path, monkeypatch):
    """``run_batch(pairs_per_gpu=2)`` / ``train_pairs``: two pairs in the same launches.  With deterministic full crops
    (use_augmentations False, min_cover 1) every pair's image and loss are bit-identical to its own ``train_model`` run."""
    from splice_amd import batch
    from splice_amd.train import train_model
    monkeypatch.setenv("SPLICE_SYNTHETIC_WEIGHTS", "1")
    over = dict(OVER, use_augmentations=False, global_A_crops_min_cover=1.0, global_B_crops_min_cover=1.0)
    q, ser = tmp_path / "grouped", tmp_path / "serial"
    for r in (q, ser):
        r.mkdir()
        _write_pairs(r, 3, 72, 72)       # square: the full-cover crop has no position draw, the run is deterministic
    res = batch.run_batch(str(q), 1, over, pairs_per_gpu=2)
    assert [r["pair"] for r in res] == ["p0", "p1", "p2"]
    assert [r.get("pairs_in_step", 1) for r in res] == [2, 2, 1]          # two ride together, the odd one runs alone
    for i, r in enumerate(res):
        eng = train_model(str(ser / f"p{i}"), cfg_overrides=over, progress=False)
        assert eng.losses()["loss"] == r["loss"], (i, eng.losses()["loss"], r["loss"])
        assert (ser / f"p{i}" / "out" / "output.png").read_bytes() == (q / f"p{i}" / "out" / "output.png").read_bytes()


def test_train_pairs_state_dict_equals_single_runs(tmp_path, monkeypatch):
    """Everything ``netG.state_dict()`` holds -- parameters, BatchNorm running statistics (incl. the logging forwards, booked in
    the reference's order) and ``num_batches_tracked`` -- comes out of ``train_pairs`` as out of each pair's ``train_model``."""
    import torch
    from splice_amd.train import train_model, train_pairs
    monkeypatch.setenv("SPLICE_SYNTHETIC_WEIGHTS", "1")
    over = dict(OVER, use_augmentations=False, global_A_crops_min_cover=1.0, global_B_crops_min_cover=1.0, n_epochs=13, log_images_freq=4)
    _write_pairs(tmp_path, 2, 72, 72)
    roots = [str(tmp_path / f"p{i}") for i in range(2)]
    both = train_pairs(roots, cfg_overrides=over, progress=False)
    for i, r in enumerate(roots):
        one = train_model(r, cfg_overrides=over, progress=False).state_dict()
        got = both.state_dict(i)
        assert list(got) == list(one)
        for k in one:
            assert torch.equal(got[k], one[k]), (i, k)
        assert int(one["1.0.2.num_batches_tracked"]) == 2 * 13 + 1 + 3   # A and B crops every step, the entire image at step 0, three logged images
