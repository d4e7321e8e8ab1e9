"""CPU: the pair queue of splice_amd/batch.py (one process per worker pulling from ONE work list; nothing shared but its head).
A stub runner stands in for the GPU engine: its output is a deterministic function of the pair's images and the
overrides, so "N workers == serial" is checked bit for bit, as is the work list, the pull order under skewed run times and
the failure reporting."""
import hashlib
import json
import os
import signal
import time

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
    if overrides.get("segv") == os.path.basename(pair_dir):   # a crash below Python, the first `segv_times` times this pair is run
        marks = [f for f in os.listdir(pair_dir) if f.startswith("crashed")]
        if len(marks) < overrides.get("segv_times", 1):
            open(os.path.join(pair_dir, f"crashed{len(marks)}"), "w").close()
            os.kill(os.getpid(), signal.SIGKILL)   # (SIGKILL: dies by a signal like a SIGSEGV, without a core file)
    if overrides.get("die_at_exit"):   # the process dies by a signal AFTER its last item (what a HIP runtime crash at interpreter teardown looks like)
        from multiprocessing import util as mp_util
        mp_util.Finalize(None, os.kill, args=(os.getpid(), signal.SIGKILL), exitpriority=-100)
    t0 = time.time()
    time.sleep(overrides.get("sleep", {}).get(os.path.basename(pair_dir), overrides.get("sleep_default", 0.0)))
    return {"digest": h.hexdigest(), "steps": overrides.get("n_epochs", 0), "pid": os.getpid(), "t0": t0, "t1": time.time()}


def stub_group_runner(pair_dirs, overrides):
    return [dict(stub_runner(d, overrides), pairs_in_step=len(pair_dirs)) for d in pair_dirs]


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
        res = batch.run_batch(str(root), n, {"n_epochs": 7, "sleep_default": 0.3}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
        roots.append((root, res))
    (r1, a), (r2, b) = roots
    assert [x["pair"] for x in a] == [f"pair{i:02d}" for i in range(5)] == [x["pair"] for x in b]
    assert [x["digest"] for x in a] == [x["digest"] for x in b]                      # result independent of N
    assert {x["gpu"] for x in a} == {0} and {x["gpu"] for x in b} == {0, 1}           # both workers pulled work
    assert len({x["pid"] for x in b}) == 2                                           # one process per worker
    for i in range(5):
        assert (r1 / f"pair{i:02d}" / "out" / "output.png").read_bytes() == (r2 / f"pair{i:02d}" / "out" / "output.png").read_bytes()
    assert len({x["digest"] for x in a}) == 5                                        # pairs differ


def test_pull_queue_does_not_straggle_behind_a_slow_pair(tmp_path):
    """unequal pairs.  pair00 takes 3 s, the other five 0.25 s each.  The static `i mod N` plan gave worker 0
    the pairs 0, 2, 4 (3.5 s) next to worker 1's 0.75 s; with the pull queue the worker that drew the slow pair runs nothing
    else and the other one drains the rest while it is still busy."""
    _make_pairs(tmp_path, 6)
    over = {"sleep": {"pair00": 3.0}, "sleep_default": 0.25}
    res = batch.run_batch(str(tmp_path), 2, over, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    slow_gpu = res[0]["gpu"]
    assert [r["gpu"] for r in res[1:]] == [1 - slow_gpu] * 5, [r["gpu"] for r in res]
    assert max(r["t1"] for r in res[1:]) < res[0]["t1"]             # the five short pairs were done before the slow one ended
    # same outputs as the serial run, whatever the order the workers drew them in
    ser = tmp_path / "serial"
    ser.mkdir()
    _make_pairs(ser, 6)
    ref = batch.run_batch(str(ser), 1, over, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    assert [r["digest"] for r in ref] == [r["digest"] for r in res]


def test_work_items_are_a_function_of_the_directory_only():
    a, b, c = ((224, 224), (224, 224)), ((448, 448), (224, 224)), ((320, 240), (224, 224))
    sizes = [a, b, a, a, c, a, b, a]
    # one pair per item: most expensive images first, ties in index order
    assert batch.work_items(sizes, 1) == [[1], [6], [4], [0], [2], [3], [5], [7]]
    # groups of up to 4 equal-size pairs first by cost (4 x a = 401 k pixels, 2 x b = 502 k), then what fills no group
    assert batch.work_items(sizes, 4) == [[1, 6], [0, 2, 3, 5], [4], [7]]
    assert batch.work_items(sizes, 2) == [[1, 6], [0, 2], [3, 5], [4], [7]]
    assert batch.work_items([a], 8) == [[0]]


def test_groups_do_not_depend_on_the_number_of_workers(tmp_path):
    """Round 3 grouped per worker (pairs i mod N), so the groups -- and with random crops the results -- changed with N.  Now the
    groups come from the whole directory."""
    sizes = [((8, 8), (8, 8))] * 5 + [((16, 8), (8, 8))]
    seen = []
    for n in (1, 2, 3):
        root = tmp_path / f"n{n}"
        root.mkdir()
        _make_pairs(root, 6)
        res = batch.run_batch(str(root), n, {"sleep_default": 0.05}, runner="test_batch_cpu:stub_runner", group_runner="test_batch_cpu:stub_group_runner",
                              pin_gpu=False, pairs_per_gpu=2, sizes=sizes)
        seen.append([(r["pair"], r.get("pairs_in_step", 1), r["digest"]) for r in res])
    assert seen[0] == seen[1] == seen[2]
    assert [k for _, k, _ in seen[0]] == [2, 2, 2, 2, 1, 1]


def test_worker_failure_is_reported(tmp_path):
    _make_pairs(tmp_path, 3)
    with pytest.raises(RuntimeError, match=r"while running \['pair01'\].*without a result: \['pair01'\]"):
        batch.run_batch(str(tmp_path), 2, {"explode": "pair01", "sleep_default": 0.3}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    for ok in ("pair00", "pair02"):                                     # the healthy worker drained the rest of the queue
        assert (tmp_path / ok / "out" / "result.json").exists()
    with pytest.raises(ValueError):
        batch.run_batch(str(tmp_path / "not_a_pair"), 1, runner="test_batch_cpu:stub_runner", pin_gpu=False)


def test_killed_worker_is_replaced_and_its_item_retried(tmp_path):
    """A worker killed by a signal (the runner kills its process: what a crash of the HIP runtime looks like to the parent) is replaced
    and the item it was running goes back to the queue -- once by default; results equal the crash-free run's.  An item that keeps
    killing its workers, and Python-level failures, still take the batch down with the pair named."""
    good, bad = tmp_path / "good", tmp_path / "bad"
    for r in (good, bad):
        r.mkdir()
        _make_pairs(r, 4)
    ref = batch.run_batch(str(good), 2, {"n_epochs": 3}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    res = batch.run_batch(str(bad), 2, {"n_epochs": 3, "segv": "pair02", "sleep_default": 0.1}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    # (the digest covers the overrides: compare the crash-free fields)
    assert [r["pair"] for r in res] == [r["pair"] for r in ref] and all(r["steps"] == 3 for r in res)
    assert (bad / "pair02" / "crashed0").exists() and not (bad / "pair02" / "crashed1").exists()
    again = tmp_path / "again"
    again.mkdir()
    _make_pairs(again, 3)
    with pytest.raises(RuntimeError, match=r"exit -9\) while running \['pair01'\].*without a result: \['pair01'\]"):
        batch.run_batch(str(again), 2, {"segv": "pair01", "segv_times": 5, "sleep_default": 0.1}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    assert len([f for f in os.listdir(again / "pair01") if f.startswith("crashed")]) == 2      # first run + one retry
    with pytest.raises(RuntimeError, match=r"exit -9"):                                          # max_retries=0: the old behaviour
        batch.run_batch(str(again), 1, {"segv": "pair00", "segv_times": 1}, runner="test_batch_cpu:stub_runner", pin_gpu=False, max_retries=0)


def test_worker_killed_after_its_last_item_is_not_a_failure(tmp_path):
    """a worker that finished every item it claimed and is then killed by a signal (teardown crash) leaves current = -1, nothing
    lost and an empty queue -- every pair has its result.json, so run_batch returns the results instead of raising."""
    _make_pairs(tmp_path, 3)
    res = batch.run_batch(str(tmp_path), 2, {"n_epochs": 2, "die_at_exit": True}, runner="test_batch_cpu:stub_runner", pin_gpu=False)
    assert [r["pair"] for r in res] == ["pair00", "pair01", "pair02"] and all(r["steps"] == 2 for r in res)


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
    assert [fp8_mode({"fp8": v}) for v in ("attention", "all", " Attention ")] == ["attention"] * 3
    with pytest.raises(ValueError):
        fp8_mode({"fp8": "e5m2"})
