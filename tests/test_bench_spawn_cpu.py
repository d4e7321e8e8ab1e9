"""CPU: bench.py's own N > 1 launcher (`python bench.py --gpus N` without torchrun), exercised without a GPU through the
SPLICE_BENCH_STUB hook (the engine is a sleep-based fake; rendezvous, gloo barrier, max-over-ranks timing, device binding and
the JSON line are the real code paths).  Also the bench's refusal of debugging / work-skipping switches."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, BENCH, *argv], env=env, capture_output=True, text=True, timeout=600)


def test_spawn_two_workers_stub():
    r = _run({"SPLICE_BENCH_STUB": "20", "SPLICE_BENCH_STUB_GPUS": "4", "HIP_VISIBLE_DEVICES": "3,5,6,7", "SPLICE_CONV_TILE": "1"},
             "--gpus", "2", "--steps", "6", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints the one line
    out = json.loads(lines[0])
    assert out["metric"] == "stub_steps_per_sec" and out["data"].startswith("stub")     # can never pass for a measurement
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 1 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["per_rank_device"] == [3, 5]     # worker i inherits the i-th entry of the parent's HIP_VISIBLE_DEVICES
    r0, r1 = cfg["per_rank_steps_per_s"]
    # rank 1's step sleeps 1.5x as long (30 ms); the stop barrier is inside the timed bracket, so BOTH ranks' clocks show the
    # slow rank and value = whole-job steps / MAX-over-ranks time
    assert 20.0 < r1 < 34.0 and 20.0 < r0 < 34.0
    assert abs(out["value"] - 2 * min(r0, r1)) / out["value"] < 0.05
    assert 29.0 < out["ms_per_step"] < 45.0
    assert cfg["host"]["threads"] >= 1
    assert cfg["env"] == {"SPLICE_CONV_TILE": "1"}   # library switches are echoed, bench plumbing is not
    # blocks of exactly K steps repeated until the minimum timed duration (0.2 s for the stub; 1 s for real runs),
    # round 5: value / ms_per_step are ALL timed steps over ALL timed seconds; the median block is a robustness figure
    t = cfg["timing"]
    assert t["steps_per_block"] == 6 and t["blocks"] >= 2 and t["timed_seconds"] >= 0.2 and len(t["ms_per_step_by_block"]) == t["blocks"]
    assert abs(t["timed_seconds"] / (t["blocks"] * t["steps_per_block"]) * 1e3 - out["ms_per_step"]) < 1e-2
    assert abs(t["ms_per_step_by_block"][t["median_block"]] - t["ms_per_step_median_block"]) < 1e-6
    assert t["ms_per_step_min"] <= out["ms_per_step"] <= t["ms_per_step_max"]


def test_spawn_fails_cleanly_without_enough_gpus():
    r = _run({"SPLICE_BENCH_STUB": "5", "SPLICE_BENCH_STUB_GPUS": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "0")
    assert r.returncode != 0
    assert "only 1 GPU(s) visible" in r.stderr + r.stdout
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _run({"SPLICE_BENCH_STUB": "5", "WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--steps", "2")
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr + r.stdout


def test_dev_switches_are_refused(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k in list(os.environ):
        if k.startswith("SPLICE_"):
            monkeypatch.delenv(k)
    assert bench.dev_env_violations(False) == []
    monkeypatch.setenv("SPLICE_CONV_TILE_MIN", "40000")            # a launch-policy knob: echoed, not refused
    assert bench.dev_env_violations(False) == [] and bench.library_env() == {"SPLICE_CONV_TILE_MIN": "40000"}
    monkeypatch.setenv("SPLICE_STEP_ABLATE", "3")
    monkeypatch.setenv("SPLICE_STEP_GRAPH", "0")
    monkeypatch.setenv("SPLICE_STEP_OVERLAP", "1")
    bad = bench.dev_env_violations(True)
    assert "SPLICE_STEP_ABLATE=3" in bad and "SPLICE_STEP_GRAPH=0" in bad and not any("OVERLAP" in b for b in bad)
    assert any("DEV=1" in b for b in bad)


def test_product_library_has_no_ablation_switch():
    """The work-skipping timing switch is compiled out of the shipped library (make DEV=1 builds a scratch copy)."""
    from splice_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    assert _lib.lib().splice_dev_switches() == 0
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"SPLICE_STEP_ABLATE" not in blob
