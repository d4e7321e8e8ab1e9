"""GPU: splice_amd.batch on the real engine -- a directory of pairs through the worker queue (N = 1 worker process on the
one GPU of the test box) writes bit-identical images / losses to running train_model on each pair in THIS process."""
import os

import numpy as np
import pytest

from splice_amd import synth

pytestmark = pytest.mark.gpu
OVER = dict(seed=3, n_epochs=12, dino_model_name="dino_vits8", dino_global_patch_size=64, log_images_freq=6)


def _write_pairs(root, k):
    from PIL import Image
    for i in range(k):
        A, B = synth.smooth_image_pair(60, i, 64, 80)
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
