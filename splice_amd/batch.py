"""Many pairs on one node: a directory of K structure/appearance pairs -> a pull queue over N GPUs.

The reference optimises one pair per process invocation (``train.py:34-49,83-89``: one ``--dataroot`` with ``A/`` and
``B/``); K pairs are K independent runs.  Pairs share nothing but the frozen ViT weights, so the
node-level driver is embarrassingly parallel: one worker PROCESS per GPU (its own HIP context, pinned with
``HIP_VISIBLE_DEVICES``), no collective, no shared state but ONE integer -- the head of the work list.  The work list is built
once, by the parent, from the whole directory (``work_items``: groups of equal-size pairs when ``pairs_per_gpu`` > 1, then the
single pairs; largest images first) and every worker PULLS the next item when it is free (round 4; rounds 1-3 assigned pair
*i* to GPU *i mod N* statically, so one slow pair -- a 448 px image among 224 px ones -- left its GPU's other pairs waiting while the
rest of the node idled).  The result of a pair does not depend on N, nor on which worker ran it, nor on the order: every run
re-seeds from its config, and the grouping is a function of the directory, not of the workers.  ``run_batch(root, 1)`` and
``run_batch(root, 8)`` write bit-identical outputs (tests/test_batch_cpu.py pins that with stub runners on CPU workers, with
skewed run times; tests/test_batch_gpu.py with the real engine on one GPU).

Layout::

    root/<pair name>/A/<image>      structure image      (what ``train_model(dataroot)`` expects as dataroot)
    root/<pair name>/B/<image>      appearance image
    root/<pair name>/out/output.png written by the run
    root/<pair name>/out/result.json {"pair", "gpu", "steps", "loss", "seconds", ...}

    python -m splice_amd.batch --root pairs/ --gpus 8 [--pairs-per-gpu P] [--n_epochs 2000] [--set key=value ...]
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
    """Static form ``i mod n_gpus`` of SURVEY.md section 8e, one index list per GPU -- what ``bench.py --gpus N`` uses for its
    synthetic pairs (equal work per pair: nothing to balance).  ``run_batch`` uses the pull queue over ``work_items`` instead."""
    return [list(range(g, n_pairs, n_gpus)) for g in range(n_gpus)]


def work_items(sizes, pairs_per_gpu=1):
    """The node's work list: a list of index lists (a group of 2 .. ``pairs_per_gpu`` equal-size pairs that ride in the same
    launches, or one pair), a function of the directory only.  Order = pull order: most expensive first (pixels of both images
    x pairs of the item: longest-processing-time-first keeps the tail of the queue short), ties by first pair index."""
    idx = list(range(len(sizes)))
    groups, singles = group_equal_sizes(idx, sizes, pairs_per_gpu) if int(pairs_per_gpu) > 1 else ([], idx)
    items = groups + [[i] for i in singles]

    def cost(item):
        (aw, ah), (bw, bh) = sizes[item[0]]
        return (aw * ah + bw * bh) * len(item)
    return sorted(items, key=lambda it: (-cost(it), it[0]))


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


def _worker(gpu, visible_id, root, names, items, head, current, runner, overrides, pin_gpu, group_runner="splice_amd.batch:train_group_runner", redo=None):
    if pin_gpu:   # must happen before the HIP runtime starts in this process
        os.environ["HIP_VISIBLE_DEVICES"] = str(visible_id)
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    run, run_group = None, None
    while True:
        with head.get_lock():             # the shared words: index of the next unclaimed work item, and the items handed back
            k = next((j for j in range(len(items)) if redo[j]), None) if redo is not None else None   # (by the parent: their worker was killed)
            if k is not None:
                redo[k] = 0
            else:
                k = head.value
                head.value = k + 1
            if k < len(items):
                current[gpu] = k          # inside the locked section: a worker killed right after its claim is still attributed
        if k >= len(items):
            break
        item = items[k]
        if len(item) > 1:                 # one MultiPairEngine for the group
            run_group = run_group or _resolve(group_runner)
            for i, res in zip(item, run_group([os.path.join(root, names[i]) for i in item], dict(overrides))):
                _write_result(root, names[i], dict(res, pair=names[i], index=i, gpu=gpu))
        else:
            run = run or _resolve(runner)
            i = item[0]
            res = dict(run(os.path.join(root, names[i]), dict(overrides)) or {})
            _write_result(root, names[i], dict(res, pair=names[i], index=i, gpu=gpu))
        current[gpu] = -1


def run_batch(root, n_gpus=1, overrides=None, runner="splice_amd.batch:train_runner", pin_gpu=True, visible_ids=None, pairs_per_gpu=1,
              group_runner=None, sizes=None, max_retries=1):
    """Optimise every pair under ``root`` on ``n_gpus`` worker processes; returns the per-pair result dicts in pair order.

    ``runner``: ``"module:function"`` (or a picklable callable) ``(pair_dir, overrides) -> dict``; the default trains the
    pair.  ``pairs_per_gpu`` > 1: up to that many pairs of equal image sizes are optimised in the same launches (1.6x the
    pairs/hr of one pair at a time at 8 pairs per GPU) through ``group_runner`` ``(pair_dirs, overrides) -> [dict per pair]``
    (default: ``train_group_runner`` = ``train_pairs``); pairs that fill no group go through ``runner``.  A custom ``runner``
    without a matching ``group_runner`` is rejected when ``pairs_per_gpu`` > 1 -- grouped pairs would otherwise silently run the
    default training.  The groups are formed over the WHOLE directory (``work_items``), so they do not depend on ``n_gpus``.
    One thing differs from K single runs in a group: its pairs share one crop SIZE per step (``PairBatchFeed``: positions and
    augmentations stay per pair), so under random crops a pair's RNG stream is not the one of its single run (with deterministic
    full crops the results are bit-identical, tests/test_batch_gpu.py).
    ``sizes``: per pair ``((A_w, A_h), (B_w, B_h))`` if already known (default: read from the image headers).
    ``pin_gpu=False`` leaves device visibility alone (CPU tests).  A worker KILLED BY A SIGNAL (a crash below Python: the HIP
    runtime's handler thread has done that) is replaced by a fresh process on the same GPU and the item it
    was running goes back to the queue, at most ``max_retries`` times per item; a worker that ends with a Python error -- a
    deterministic failure -- or an item out of retries takes the batch down with a RuntimeError naming the pairs it was running and
    the pairs left undone; the other workers drain the rest of the queue first, and finished pairs keep their ``result.json``."""
    import multiprocessing as mp
    if int(pairs_per_gpu) > 1 and group_runner is None:
        if runner != "splice_amd.batch:train_runner":
            raise ValueError("run_batch: pairs_per_gpu > 1 with a custom runner needs a group_runner (pair_dirs, overrides) -> [dict per pair]")
        group_runner = "splice_amd.batch:train_group_runner"
    group_runner = group_runner or "splice_amd.batch:train_group_runner"
    names = discover_pairs(root)
    if not names:
        raise ValueError(f"{root}: no <pair>/A + <pair>/B directories found")
    if len((overrides or {}).get("dino_global_scales") or []) > 1 and int(pairs_per_gpu) > 1:
        raise ValueError("run_batch: dino_global_scales with several entries is a single-pair option; use pairs_per_gpu=1")
    if sizes is None:
        try:
            sizes = [_image_sizes(os.path.join(root, n)) for n in names]
        except Exception:
            if int(pairs_per_gpu) > 1 or runner == "splice_amd.batch:train_runner":
                raise                                  # the default runner trains IMAGES: an unreadable one is an error here, not later in a worker
            sizes = [((0, 0), (0, 0))] * len(names)    # custom (stub) runners on directories without images: index order, one pair per item
    items = work_items(sizes, pairs_per_gpu)
    n_gpus = max(1, min(int(n_gpus), len(items)))
    if visible_ids is None:
        parent = os.environ.get("HIP_VISIBLE_DEVICES")
        visible_ids = parent.split(",") if parent else [str(g) for g in range(n_gpus)]
    if pin_gpu and len(visible_ids) < n_gpus:
        raise ValueError(f"run_batch: {n_gpus} workers requested, {len(visible_ids)} visible GPUs")
    for name in names:                      # a stale result of an earlier batch must not pass for this one's
        try:
            os.remove(os.path.join(root, name, "out", "result.json"))
        except OSError:
            pass
    ctx = mp.get_context("spawn")   # fresh interpreters: the HIP runtime must not be inherited through fork
    head = ctx.Value("i", 0)
    current = ctx.Array("i", [-1] * n_gpus)
    redo = ctx.Array("i", [0] * len(items), lock=False)   # (guarded by head's lock)

    def spawn(g):
        p = ctx.Process(target=_worker, args=(g, visible_ids[g] if pin_gpu else g, root, names, items, head, current, runner, dict(overrides or {}), pin_gpu, group_runner, redo))
        p.start()
        return p

    alive = {g: spawn(g) for g in range(n_gpus)}
    attempts, failures = [0] * len(items), []
    while alive:
        for g, p in list(alive.items()):
            p.join(timeout=0.05)
            if p.exitcode is None:
                continue
            del alive[g]
            if p.exitcode == 0:
                continue
            k = current[g]
            if p.exitcode < 0:
                # a worker killed INSIDE the claim section leaves the shared lock held: every other worker would block on it for ever
                lk = head.get_lock()
                if lk.acquire(timeout=10.0):
                    lk.release()
                else:
                    for q in alive.values():
                        q.terminate()
                    undone = [n for n in names if not os.path.exists(os.path.join(root, n, "out", "result.json"))]
                    raise RuntimeError(f"run_batch: worker of gpu {g} was killed by signal {-p.exitcode} while holding the work-list lock; "
                                       f"the batch cannot continue; pairs without a result: {undone}")
                if k < 0:
                    # killed between two items (or before its first claim was visible): hand back whatever was claimed, is unfinished and is not
                    # some live worker's current item
                    with lk:
                        claimed = min(head.value, len(items))
                        busy = {current[h] for h in alive}
                        lost = [j for j in range(claimed) if not redo[j] and j not in busy and attempts[j] < int(max_retries)
                                and any(not os.path.exists(os.path.join(root, names[i], "out", "result.json")) for i in items[j])]
                        for j in lost:
                            redo[j] = 1
                            attempts[j] += 1
                    if lost or head.value < len(items):
                        print(f"run_batch: worker of gpu {g} was killed by signal {-p.exitcode} between items; re-queued {[[names[i] for i in items[j]] for j in lost]}, "
                              "restarting it", file=sys.stderr, flush=True)
                        alive[g] = spawn(g)
                        continue
                    # nothing lost and the queue is drained: the worker had finished its last item and died afterwards (a HIP runtime crash at
                    # interpreter teardown is exactly this case) -- every pair it claimed has its result.json, so this is not a failure
                    with lk:
                        busy = {current[h] for h in alive}
                        orphaned = [j for j in range(min(head.value, len(items))) if not redo[j] and j not in busy
                                    and any(not os.path.exists(os.path.join(root, names[i], "out", "result.json")) for i in items[j])]
                    if not orphaned:
                        print(f"run_batch: worker of gpu {g} was killed by signal {-p.exitcode} after its last item; all results are on disk", file=sys.stderr, flush=True)
                        continue
            if p.exitcode < 0 and k >= 0 and attempts[k] < int(max_retries):   # killed by a signal while running item k: hand it back, new worker
                attempts[k] += 1
                print(f"run_batch: worker of gpu {g} was killed by signal {-p.exitcode} while running {[names[i] for i in items[k]]}; "
                      f"restarting it (retry {attempts[k]} of {int(max_retries)} for that item)", file=sys.stderr, flush=True)
                with head.get_lock():
                    redo[k] = 1
                current[g] = -1
                alive[g] = spawn(g)
            else:
                failures.append((g, p.exitcode, [names[i] for i in items[k]] if k >= 0 else []))
    if failures:
        undone = [n for n in names if not os.path.exists(os.path.join(root, n, "out", "result.json"))]
        raise RuntimeError("run_batch: worker(s) failed: " + "; ".join(f"gpu {g} (exit {code}) while running {running}" for g, code, running in failures)
                           + f"; pairs without a result: {undone}")
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
    ap = ArgumentParser(description="Optimise a directory of Splice pairs over the GPUs of one node (one worker per GPU pulling from one work list).")
    ap.add_argument("--root", required=True, help="directory of <pair>/A, <pair>/B sub-directories")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--pairs-per-gpu", type=int, default=1, help="pairs of equal image sizes optimised in the same launches on a GPU")
    ap.add_argument("--n_epochs", type=int, default=None, help="optimisation steps per pair (config default otherwise)")
    ap.add_argument("--scales", default="", help="comma list of ViT input sizes evaluated per step, e.g. 224,320,448 (config key dino_global_scales; BASELINE configs[4])")
    ap.add_argument("--fp8", nargs="?", const="gemm", default=None, choices=("gemm", "attention"),
                    help="e4m3 operands for the QKV / fc1 / fc2 projections and the self-similarity Gram matrices (config key fp8); "
                         "'--fp8 attention': the attention forward too")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="config override (conf/default/config.yaml keys)")
    ap.add_argument("--max-retries", type=int, default=1, help="times an item goes back to the queue when the worker running it is killed by a signal")
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
    res = run_batch(args.root, args.gpus, over, pairs_per_gpu=args.pairs_per_gpu, max_retries=args.max_retries)
    dt = time.perf_counter() - t0
    print(json.dumps({"pairs": len(res), "gpus": args.gpus, "seconds": round(dt, 2), "pairs_per_hour": round(len(res) * 3600 / dt, 2), "results": res}))


if __name__ == "__main__":
    sys.exit(main())
