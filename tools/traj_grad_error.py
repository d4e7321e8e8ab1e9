#!/usr/bin/env python3
"""Why does the engine's gradient error against the fp32 oracle grow along the free run (VERDICT r5 "What's weak" #2)?

The free run of tests/test_step_gpu.py::test_trajectory_a (64 x 64 pair, ViT-S/8, 78 steps).  At every spot step the fp32 oracle is
evaluated at the engine's own parameters and, for the engine's gradient g_e and the oracle's g_o (whole arena, the analytically-zero
conv biases excluded):

    |g_o|, |g_e - g_o| (absolute), the ratio, cos(g_e, g_o), the component of the error ALONG g_o (a bias in the step length)
    and -- the test for "isotropic noise, not a bias" -- the cosine between the error vectors of DIFFERENT spot steps
    (a systematic error direction would show up as a large |cos|; independent rounding noise gives ~ 1/sqrt(n_eff)).

Also per spot the same comparison with the three loss terms switched on one at a time (lambda of the others = 0) at the SAME parameters,
which says which term the error comes from.  Run on the GPU box; prints a table (copy into profiles/)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import dino_vit, losses as OL
from oracle.step import SpliceOracle
from splice_amd import synth
from splice_amd.engine import SpliceEngine

DEV = "cuda"
SPOTS = [1, 6, 12, 18, 24, 30, 36, 42, 48, 54, 60, 66, 72, 77]


def oracle_for(cfg, vit_state, round_weights=False):
    patch, dim, depth, heads = dino_vit.DINO_CONFIGS["dino_vits8"]
    m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=64).eval()
    if round_weights:
        m.load_state_dict(dino_vit.round_weights_bf16(vit_state, dim))
        return SpliceOracle(m, {k: torch.from_numpy(v) for k, v in synth.generator_params(1, 0.02).items()}, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vit_state.items()})
    return SpliceOracle(m, {k: torch.from_numpy(v) for k, v in synth.generator_params(1, 0.02).items()}, cfg)


def flat_pair(eng, og):
    ge, go = [], []
    for (name, gt), o in zip(eng.gen.unflatten(eng.grads).items(), og):
        if name.endswith("0.bias") and name != "9.0.bias":
            continue
        ge.append(gt.detach().cpu().double().reshape(-1))
        go.append(o.detach().double().reshape(-1))
    return torch.cat(ge), torch.cat(go)


def eval_at(snap, step, cfg_over, A, B, vit_state):
    """engine and oracle gradient at the parameters `snap`, step index `step`, with config overrides (lambdas)"""
    # (a fresh step handle switches the ssim / identity terms on when it is handed step index == cls_warmup: make that THIS step)
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=64, cls_warmup=step, **cfg_over)
    eng = SpliceEngine(cfg, vit_state, synth.generator_params(31, 0.02), A.shape[-2:], A.shape[-2:])
    eng.params.copy_(eng.gen.flatten({k: v for k, v in snap.items()}))
    eng.step_idx = step - 1
    orc = oracle_for(eng.cfg, vit_state)
    with torch.no_grad():
        for k, v in orc.params.items():
            v.copy_(snap[k].cpu().reshape(v.shape))
    orc.step_idx = step - 1
    orc.lambdas = OL.initial_lambdas(orc.cfg)
    if step >= orc.cfg["cls_warmup"]:
        OL.update_lambdas(orc.lambdas, orc.cfg, orc.cfg["cls_warmup"])
    return eng, orc


def main():
    A, B = synth.smooth_image_pair(32, 0, 64, 64)
    cfg = dict(dino_model_name="dino_vits8", dino_global_patch_size=64)
    vit_state = synth.vit_params(7, "dino_vits8", img_size=64, w_std=0.05)
    eng = SpliceEngine(cfg, vit_state, synth.generator_params(31, 0.02), A.shape[-2:], A.shape[-2:])
    orc = oracle_for(eng.cfg, vit_state)
    orc_w = oracle_for(eng.cfg, vit_state, round_weights=True)   # the fp32 oracle on the bf16-rounded WEIGHTS the engine computes with
    At, Bt = torch.from_numpy(A), torch.from_numpy(B)
    Ad, Bd = At.to(DEV), Bt.to(DEV)
    errs, errs_w, rows, rows_w = {}, {}, [], []
    print("step   loss    |g_o|      |g_e-g_o|   rel      cos(g_e,g_o)  err_along_g_o/|g_o|   per-term rel err (ssim | cls | id) and their |g_o| share")
    for step in range(78):
        if step in SPOTS:
            snap = {k: v.clone() for k, v in eng.gen.unflatten(eng.params.clone()).items()}
        eng.step(Ad, Bd, Ad)
        if step not in SPOTS:
            continue
        with torch.no_grad():
            for k, v in orc.params.items():
                v.copy_(snap[k].cpu().reshape(v.shape))
        orc.step_idx = step - 1
        orc.lambdas = OL.initial_lambdas(orc.cfg)
        if step >= orc.cfg["cls_warmup"]:
            OL.update_lambdas(orc.lambdas, orc.cfg, orc.cfg["cls_warmup"])
        lo, _, og = orc.step(At[None], Bt[None], At[None])
        with torch.no_grad():
            for k, v in orc_w.params.items():
                v.copy_(snap[k].cpu().reshape(v.shape))
        orc_w.step_idx = step - 1
        orc_w.lambdas = OL.initial_lambdas(orc_w.cfg)
        if step >= orc_w.cfg["cls_warmup"]:
            OL.update_lambdas(orc_w.lambdas, orc_w.cfg, orc_w.cfg["cls_warmup"])
        _, _, ogw = orc_w.step(At[None], Bt[None], At[None])
        _, gow = flat_pair(eng, ogw)
        ge, go = flat_pair(eng, og)
        d = ge - go
        errs[step] = d
        errs_w[step] = ge - gow
        rows_w.append((step, gow.norm().item(), (ge - gow).norm().item(), (go - gow).norm().item()))
        n_o, n_d = go.norm().item(), d.norm().item()
        cos = (ge @ go).item() / (ge.norm().item() * n_o)
        along = (d @ go).item() / (n_o * n_o)
        # one loss term at a time, same parameters
        per = []
        for key in (() if os.environ.get("TRAJ_NO_TERMS") else ("lambda_global_ssim", "lambda_global_cls", "lambda_global_identity")):
            over = {k: 0.0 for k in ("lambda_global_ssim", "lambda_global_cls", "lambda_global_identity", "lambda_entire_ssim", "lambda_entire_cls")}
            over[key] = eng.cfg[key]
            e1, o1 = eval_at(snap, step, over, A, B, vit_state)
            e1.step(Ad, Bd, Ad)
            l1, _, og1 = o1.step(At[None], Bt[None], At[None])
            g1e, g1o = flat_pair(e1, og1)
            per.append(((g1e - g1o).norm().item(), g1o.norm().item()))
            del e1, o1
        rows.append((step, lo["loss"], n_o, n_d))
        pt = " | ".join(f"{a / max(b, 1e-30):.2e} ({b / n_o:.2f})" for a, b in per) if per else "-"
        print(f"{step:4d} {lo['loss']:8.2f} {n_o:10.3e} {n_d:10.3e} {n_d / n_o:9.2e}   {cos:.6f}     {along:+.2e}          {pt}", flush=True)
    # error vectors of different steps against each other
    ks = sorted(errs)
    C = np.zeros((len(ks), len(ks)))
    for i, a in enumerate(ks):
        for j, b in enumerate(ks):
            C[i, j] = (errs[a] @ errs[b]).item() / (errs[a].norm().item() * errs[b].norm().item())
    off = C[~np.eye(len(ks), dtype=bool)]
    print("cosine between the error vectors of different spot steps: mean %+.3f, max |cos| %.3f (n = %d parameters)" % (off.mean(), np.abs(off).max(), errs[ks[0]].numel()))
    print("adjacent spots:", " ".join(f"{C[i, i + 1]:+.3f}" for i in range(len(ks) - 1)))
    print("\nagainst the fp32 oracle evaluated on the bf16-ROUNDED WEIGHTS the engine computes with (what is left is the rounding of activations / probabilities):")
    print("step   |g_w|      |g_e-g_w|   rel      |g_o-g_w| (the weight rounding alone)  rel")
    for (st_, nw, de, dw) in rows_w:
        print(f"{st_:4d} {nw:10.3e} {de:10.3e} {de / nw:9.2e}   {dw:10.3e} {dw / nw:9.2e}")
    Cw = np.zeros((len(ks), len(ks)))
    for i, a in enumerate(ks):
        for j, b in enumerate(ks):
            Cw[i, j] = (errs_w[a] @ errs_w[b]).item() / (errs_w[a].norm().item() * errs_w[b].norm().item())
    offw = Cw[~np.eye(len(ks), dtype=bool)]
    print("cosine between those error vectors of different spot steps: mean %+.3f, max |cos| %.3f; adjacent: %s" % (offw.mean(), np.abs(offw).max(), " ".join(f"{Cw[i, i + 1]:+.3f}" for i in range(len(ks) - 1))))
    r = np.array(rows)
    print("log-log slope of |g_e - g_o| against |g_o| over the spots: %.2f (1 = proportional error, 0 = fixed floor)" % np.polyfit(np.log(r[:, 2]), np.log(r[:, 3]), 1)[0])


if __name__ == "__main__":
    main()
