"""CPU: the pair queue of splice_amd/batch.py (pair i -> worker i mod N, one process per worker, no shared state).
A stub runner stands in for the GPU engine: its output is a deterministic function of the pair's images and the
overrides, so "N workers == serial" is checked bit for bit, as is the order / assignment / failure reporting."""
import hashlib
import json
import os

import numpy as np
import pytest

from splice_amd import batch, synth


def stub_runner(pair_dir, overrides):
    """Deterministic stand-in for train_model: digest of both images + the overrides; also writes out/output.png bytes."""
    h = hashlib.sha256(json.dumps(overrides, sort_keys=True).encode())
    for side in ("A", "B"):
        d = os.path.join(pair_dir, side)
        with open(os.path.join(d, sorted(os.listdir(d))[0]), "rb") as f:
            h.update(f.read())
    os.makedirs(os.path.join(pair_dir, "out"), exist_ok=True)
    with open(os.path.join(pair_dir, "out", "output.png"), "wb") as f:
        f.write(h.digest())
    if overrides.get("explode") == os.path.basename(pair_dir):
        raise RuntimeError("boom")
    return {"digest": h.hexdigest(), "steps": overrides.get("n_epochs", 0), "pid": os.getpid()}


def _make_pairs(root, k):
    for i in range(k):
        A, B = synth.image_pair(99, i, 8, 8)
        for side, img in (("A", A), ("B", B)):
            d = root / f"pair{i:02d}" / side
            d.mkdir(parents=True)
            np.save(d / "img.npy", img)
    (root / "not_a_pair").mkdir()
    (root / "half" / "A").mkdir(parents=True)


def test_assignment_round_robin():
    assert batch.assignment(5, 2) == [[0, 2, 4], [1, 3]]
    assert batch.assignment(3, 8)[:3] == [[0], [1], [2]]


def test_queue_matches_serial(tmp_path):
    roots = []
    for tag, n in (("serial", 1), ("two", 2)):
        root = tmp_path / tag
        root.mkdir()
        _make_pairs(root, 5)
        res = batch.run_batch(str(root), n, {"n_epochs": 7}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
        roots.append((root, res))
    (r1, a), (r2, b) = roots
    assert [x["pair"] for x in a] == [f"pair{i:02d}" for i in range(5)] == [x["pair"] for x in b]
    assert [x["digest"] for x in a] == [x["digest"] for x in b]                      # result independent of N
    assert [x["gpu"] for x in b] == [0, 1, 0, 1, 0] and {x["gpu"] for x in a} == {0}   # pair i -> worker i mod N
    assert len({x["pid"] for x in b}) == 2                                           # one process per worker
    for i in range(5):
        assert (r1 / f"pair{i:02d}" / "out" / "output.png").read_bytes() == (r2 / f"pair{i:02d}" / "out" / "output.png").read_bytes()
    assert len({x["digest"] for x in a}) == 5                                        # pairs differ


def test_worker_failure_is_reported(tmp_path):
    _make_pairs(tmp_path, 3)
    with pytest.raises(RuntimeError, match="gpu 1"):
        batch.run_batch(str(tmp_path), 2, {"explode": "pair01"}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    assert (tmp_path / "pair00" / "out" / "result.json").exists()      # the healthy worker finished its pairs
    with pytest.raises(ValueError):
        batch.run_batch(str(tmp_path / "not_a_pair"), 1, runner="test_batch_cpu:stub_runner", pin_gpu=False)


def test_group_equal_sizes():
    from splice_amd.batch import group_equal_sizes
    a, b = ((224, 224), (224, 224)), ((320, 240), (224, 224))
    # five pairs of size a, two of size b, one odd size; groups of up to 4
    idx = [0, 1, 2, 3, 4, 5, 6, 7]
    sizes = [a, b, a, a, ((1, 1), (1, 1)), a, b, a]
    groups, singles = group_equal_sizes(idx, sizes, 4)
    assert groups == [[0, 2, 3, 5], [1, 6]] and singles == [4, 7]
    groups, singles = group_equal_sizes(idx, sizes, 2)
    assert groups == [[0, 2], [1, 6], [3, 5]] and singles == [4, 7]
    assert group_equal_sizes([3], [a], 8) == ([], [3])


def test_custom_runner_needs_a_group_runner_when_pairs_share_launches(tmp_path):
    """with pairs_per_gpu > 1 the grouped pairs go through the GROUP runner; a custom runner alone is rejected
    (it used to be ignored silently for them)."""
    _make_pairs(tmp_path, 2)
    with pytest.raises(ValueError, match="group_runner"):
        batch.run_batch(str(tmp_path), 1, runner="test_batch_cpu:stub_runner", pin_gpu=False, pairs_per_gpu=2)


def test_fp8_config_key_maps_to_the_engine_mode():
    """``fp8: True`` selects the fastest measured e4m3 setting (projections + Gram: engine mode "gemm"); the attention forward in
    e4m3 is asked for by name."""
    from splice_amd.train import fp8_mode
    assert [fp8_mode({"fp8": v}) for v in (False, None, 0, "off", "False")] == [False] * 5
    assert fp8_mode({}) is False
    assert [fp8_mode({"fp8": v}) for v in (True, 1, "gemm", "True")] == ["gemm"] * 4
    assert [fp8_mode({"fp8": v}) for v in ("attention", "all", " Attention ")] == [True] * 3
