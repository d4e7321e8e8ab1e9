"""Many pairs on one node: a directory of K structure/appearance pairs -> a queue over N GPUs.

The reference optimises one pair per process invocation (``train.py:34-49,83-89``: one ``--dataroot`` with ``A/`` and
``B/``); K pairs are K independent runs.  Pairs share nothing but the frozen ViT weights, so the
node-level driver is embarrassingly parallel: pair *i* goes to GPU *i mod N*, one worker PROCESS per GPU (its own HIP
context, pinned with ``HIP_VISIBLE_DEVICES``), each worker walks its pairs in index order, no collective, no shared
state.  The result of a pair therefore does not depend on N: ``run_batch(root, 1)`` and ``run_batch(root, 8)`` write
bit-identical outputs (tests/test_batch_cpu.py pins that with a stub runner on CPU workers, tests/test_batch_gpu.py
with the real engine on one GPU).

Layout::

    root/<pair name>/A/<image>      structure image      (what ``train_model(dataroot)`` expects as dataroot)
    root/<pair name>/B/<image>      appearance image
    root/<pair name>/out/output.png written by the run
    root/<pair name>/out/result.json {"pair", "gpu", "steps", "loss", "seconds", ...}

    python -m splice_amd.batch --root pairs/ --gpus 8 [--n_epochs 2000] [--set key=value ...]
"""
import importlib
import json
import os
import sys
import time
from argparse import ArgumentParser


def discover_pairs(root):
    """Sorted names of the sub-directories of ``root`` that hold both ``A/`` and ``B/`` with at least one file each."""
    names = []
    for name in sorted(os.listdir(root)):
        d = os.path.join(root, name)
        if all(os.path.isdir(os.path.join(d, s)) and os.listdir(os.path.join(d, s)) for s in ("A", "B")):
            names.append(name)
    return names


def assignment(n_pairs, n_gpus):
    """pair index -> GPU: ``i mod n_gpus``; returned as one index list per GPU."""
    return [list(range(g, n_pairs, n_gpus)) for g in range(n_gpus)]


def train_runner(pair_dir, overrides):
    """Default runner: the drop-in ``train_model`` on the pair's directory."""
    from .train import train_model
    t0 = time.perf_counter()
    eng = train_model(pair_dir, cfg_overrides=overrides, progress=False)
    import torch
    torch.cuda.synchronize()
    return {"steps": eng.step_idx + 1, "loss": eng.losses()["loss"], "seconds": round(time.perf_counter() - t0, 3)}


def train_group_runner(pair_dirs, overrides):
    """Several pairs of one worker in the SAME launches (``train_pairs``): used when ``run_batch(pairs_per_gpu > 1)``."""
    from .train import train_pairs
    t0 = time.perf_counter()
    eng = train_pairs(pair_dirs, cfg_overrides=overrides, progress=False)
    import torch
    torch.cuda.synchronize()
    dt = round(time.perf_counter() - t0, 3)
    return [{"steps": eng.step_idx + 1, "loss": d["loss"], "seconds": dt, "pairs_in_step": len(pair_dirs)} for d in eng.losses()]


def _image_sizes(pair_dir):
    from PIL import Image
    out = []
    for side in ("A", "B"):
        d = os.path.join(pair_dir, side)
        with Image.open(os.path.join(d, sorted(os.listdir(d))[0])) as im:
            out.append(im.size)
    return tuple(out)


def group_equal_sizes(indices, sizes, pairs_per_gpu):
    """A worker's pairs -> (groups, singles): groups of 2 .. ``pairs_per_gpu`` pairs with identical image sizes (index order
    inside a group, groups in order of their first pair), and what does not fill a group, in index order."""
    by_size = {}
    for i, sz in zip(indices, sizes):
        by_size.setdefault(sz, []).append(i)
    groups, singles = [], []
    for idx in by_size.values():
        while len(idx) >= 2:
            groups.append(idx[:pairs_per_gpu])
            idx = idx[pairs_per_gpu:]
        singles += idx
    return sorted(groups, key=lambda g: g[0]), sorted(singles)


def _resolve(runner):
    if callable(runner):
        return runner
    mod, _, fn = runner.partition(":")
    return getattr(importlib.import_module(mod), fn)


def _write_result(root, name, res):
    pair_dir = os.path.join(root, name)
    os.makedirs(os.path.join(pair_dir, "out"), exist_ok=True)
    tmp = os.path.join(pair_dir, "out", "result.json.tmp")
    with open(tmp, "w") as f:
        json.dump(res, f)
    os.replace(tmp, os.path.join(pair_dir, "out", "result.json"))


def _worker(gpu, visible_id, root, names, indices, runner, overrides, pin_gpu, pairs_per_gpu=1, group_runner="splice_amd.batch:train_group_runner"):
    if pin_gpu:   # must happen before the HIP runtime starts in this process
        os.environ["HIP_VISIBLE_DEVICES"] = str(visible_id)
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    groups, todo = group_equal_sizes(indices, [_image_sizes(os.path.join(root, names[i])) for i in indices], pairs_per_gpu) if pairs_per_gpu > 1 else ([], list(indices))
    run_group = _resolve(group_runner)
    for grp in groups:   # one MultiPairEngine per group
        for i, res in zip(grp, run_group([os.path.join(root, names[i]) for i in grp], dict(overrides))):
            _write_result(root, names[i], dict(res, pair=names[i], index=i, gpu=gpu))
    run = _resolve(runner)
    for i in todo:
        res = dict(run(os.path.join(root, names[i]), dict(overrides)) or {})
        _write_result(root, names[i], dict(res, pair=names[i], index=i, gpu=gpu))


def run_batch(root, n_gpus=1, overrides=None, runner="splice_amd.batch:train_runner", pin_gpu=True, visible_ids=None, pairs_per_gpu=1,
              group_runner=None):
    """Optimise every pair under ``root`` on ``n_gpus`` worker processes; returns the per-pair result dicts in pair order.

    ``runner``: ``"module:function"`` (or a picklable callable) ``(pair_dir, overrides) -> dict``; the default trains the
    pair.  ``pairs_per_gpu`` > 1: a worker optimises up to that many of ITS pairs in the same launches (pairs of equal image
    sizes only; 1.6x the pairs/hr of one pair at a time at 8 pairs per GPU) through ``group_runner``
    ``(pair_dirs, overrides) -> [dict per pair]`` (default: ``train_group_runner`` = ``train_pairs``); pairs that fill no group
    go through ``runner``.  A custom ``runner`` without a matching ``group_runner`` is rejected when ``pairs_per_gpu`` > 1 --
    grouped pairs would otherwise silently run the default training.  Two things differ from K single runs in a group: the
    pairs of a group share one crop SIZE per step (``PairBatchFeed``: positions and augmentations stay per pair), so under
    random crops a pair's RNG stream is not the one of its single run (with deterministic full crops the results are
    bit-identical, tests/test_batch_gpu.py); and "the result does not depend on N" holds for the grouping, which is per worker.
    ``pin_gpu=False`` leaves device visibility alone (CPU tests).  A worker that dies takes the batch down with a RuntimeError
    naming its pairs; finished pairs keep their ``result.json``."""
    import multiprocessing as mp
    if int(pairs_per_gpu) > 1 and group_runner is None:
        if runner != "splice_amd.batch:train_runner":
            raise ValueError("run_batch: pairs_per_gpu > 1 with a custom runner needs a group_runner (pair_dirs, overrides) -> [dict per pair]")
        group_runner = "splice_amd.batch:train_group_runner"
    group_runner = group_runner or "splice_amd.batch:train_group_runner"
    names = discover_pairs(root)
    if not names:
        raise ValueError(f"{root}: no <pair>/A + <pair>/B directories found")
    n_gpus = max(1, min(int(n_gpus), len(names)))
    plan = assignment(len(names), n_gpus)
    if visible_ids is None:
        parent = os.environ.get("HIP_VISIBLE_DEVICES")
        visible_ids = parent.split(",") if parent else [str(g) for g in range(n_gpus)]
    if pin_gpu and len(visible_ids) < n_gpus:
        raise ValueError(f"run_batch: {n_gpus} workers requested, {len(visible_ids)} visible GPUs")
    ctx = mp.get_context("spawn")   # fresh interpreters: the HIP runtime must not be inherited through fork
    procs = [ctx.Process(target=_worker, args=(g, visible_ids[g] if pin_gpu else g, root, names, plan[g], runner, dict(overrides or {}), pin_gpu, int(pairs_per_gpu), group_runner))
             for g in range(n_gpus)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    failed = [g for g, p in enumerate(procs) if p.exitcode != 0]
    if failed:
        raise RuntimeError("run_batch: worker(s) failed: " + "; ".join(f"gpu {g} (pairs {[names[i] for i in plan[g]]}, exit {procs[g].exitcode})" for g in failed))
    out = []
    for name in names:
        with open(os.path.join(root, name, "out", "result.json")) as f:
            out.append(json.load(f))
    return out


def _parse_value(text):
    try:
        return json.loads(text)
    except ValueError:
        return text


def main(argv=None):
    ap = ArgumentParser(description="Optimise a directory of Splice pairs over the GPUs of one node (pair i -> GPU i mod N).")
    ap.add_argument("--root", required=True, help="directory of <pair>/A, <pair>/B sub-directories")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--pairs-per-gpu", type=int, default=1, help="pairs of equal image sizes optimised in the same launches on a GPU")
    ap.add_argument("--n_epochs", type=int, default=None, help="optimisation steps per pair (config default otherwise)")
    ap.add_argument("--scales", default="", help="comma list of ViT input sizes evaluated per step, e.g. 224,320,448 (config key dino_global_scales; BASELINE configs[4])")
    ap.add_argument("--fp8", nargs="?", const="gemm", default=None, choices=("gemm", "attention"),
                    help="e4m3 operands for the QKV / fc1 / fc2 projections and the self-similarity Gram matrices (config key fp8); "
                         "'--fp8 attention': the attention forward too")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="config override (conf/default/config.yaml keys)")
    args = ap.parse_args(argv)
    over = {}
    if args.n_epochs is not None:
        over["n_epochs"] = args.n_epochs
    if args.scales:
        over["dino_global_scales"] = [int(x) for x in args.scales.split(",")]
        if args.pairs_per_gpu > 1:
            raise SystemExit("--scales is a single-pair option (train_model); use --pairs-per-gpu 1")
    if args.fp8:
        over["fp8"] = args.fp8
    for kv in args.set:
        k, _, v = kv.partition("=")
        over[k] = _parse_value(v)
    t0 = time.perf_counter()
    res = run_batch(args.root, args.gpus, over, pairs_per_gpu=args.pairs_per_gpu)
    dt = time.perf_counter() - t0
    print(json.dumps({"pairs": len(res), "gpus": args.gpus, "seconds": round(dt, 2), "pairs_per_hour": round(len(res) * 3600 / dt, 2), "results": res}))


if __name__ == "__main__":
    sys.exit(main())
