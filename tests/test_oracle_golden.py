"""CPU: the oracle (our restatement) replays the fixtures recorded from the REFERENCE's
own modules (oracle/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import dino_vit, extractor as oex, generator as ogen, losses as olosses
from oracle.step import SpliceOracle
from splice_amd import synth



def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _vit(model_name, img_size, seed=7, w_std=0.05):
    patch, dim, depth, heads = dino_vit.DINO_CONFIGS[model_name]
    m = dino_vit.VisionTransformer(patch, dim, depth, heads, img_size=img_size).eval()
    sd = synth.vit_params(seed, model_name, img_size=img_size, w_std=w_std)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


def test_attn_cosine_sim(golden_dir):
    g = _load(golden_dir, "extractor.npz")
    for T, D in ((5, 8), (197, 64)):
        x = torch.from_numpy(synth.normal(11, f"cos/{T}", (1, 1, T, D)))
        np.testing.assert_allclose(oex.attn_cosine_sim(x).numpy(), g[f"cos_T{T}_D{D}"], rtol=0, atol=1e-6)
    x = torch.from_numpy(synth.normal(11, "cos/zero", (1, 1, 6, 8)).copy())
    x[0, 0, 2] = 0
    np.testing.assert_allclose(oex.attn_cosine_sim(x).numpy(), g["cos_zero_row"], rtol=0, atol=1e-6)


def test_qkv_split(golden_dir):
    g = _load(golden_dir, "extractor.npz")
    qkv = torch.from_numpy(synth.normal(12, "qkv", (1, 17, 3 * 384)))
    q, k, v = oex.split_qkv(qkv, 6)
    assert np.array_equal(q.numpy(), g["q_from_qkv"])
    assert np.array_equal(k.numpy(), g["k_from_qkv"])
    assert np.array_equal(v.numpy(), g["v_from_qkv"])
    # K10 of SURVEY.md: keys with heads concatenated == columns [D, 2D) of the raw qkv
    kc = k.transpose(0, 1).reshape(17, 384)
    assert np.array_equal(kc.numpy(), qkv[0, :, 384:768].numpy())


def test_vit_features(golden_dir):
    g = _load(golden_dir, "extractor.npz")
    vit = _vit("dino_vits8", 32)
    img = torch.from_numpy(synth.normal(13, "img32", (1, 3, 32, 32)))
    with torch.no_grad():
        f = dino_vit.forward_features(vit, img)
        np.testing.assert_allclose(f["block"][-1].numpy(), g["vits8_block_last"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(f["block"][0].numpy(), g["vits8_block0"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(f["qkv"][11].numpy(), g["vits8_qkv11"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(f["attn"][11].numpy(), g["vits8_attn11"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(oex.keys_from_input(vit, img).numpy(), g["vits8_keys11"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(oex.keys_self_sim_from_input(vit, img).numpy(), g["vits8_selfsim11"], rtol=0, atol=1e-5)
        img2 = torch.from_numpy(synth.normal(13, "img32x48", (1, 3, 32, 48)))
        np.testing.assert_allclose(oex.keys_self_sim_from_input(vit, img2).numpy(), g["vits8_32x48_selfsim11"], rtol=0, atol=1e-5)


def _gen_loss(params, tag, h, w):
    x = torch.from_numpy(synth.uniform(22, "gin/" + tag, (1, 3, h, w)))
    y = ogen.forward(params, x)
    wgt = torch.from_numpy(synth.normal(23, "gw/" + tag, (1, 3, h, w)))
    return y, (y * wgt).sum() / y.numel() + (y * y).mean()


def _stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


@pytest.mark.parametrize("tag,h,w", [("64x64", 64, 64), ("213x213", 213, 213), ("96x130", 96, 130)])
def test_generator(golden_dir, tag, h, w):
    g = _load(golden_dir, "generator.npz")
    raw = synth.generator_params(21, 0.02, perturb_bias=0.05)
    assert list(raw.keys()) == [str(s) for s in g["param_names"]]
    assert list(raw.keys()) == [n for n, _, _ in ogen.param_specs()]
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in raw.items()}
    y, loss = _gen_loss(params, tag, h, w)
    assert y.shape == (1, 3, h, w)
    np.testing.assert_allclose(loss.item(), g[f"{tag}/loss"], rtol=1e-6)
    np.testing.assert_allclose(_stats(y), g[f"{tag}/out_stats"], rtol=1e-6)
    if tag == "64x64":
        np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}/out_full"], rtol=0, atol=2e-6)
    grads = torch.autograd.grad(loss, list(params.values()))
    got = np.stack([_stats(x) for x in grads])
    np.testing.assert_allclose(got[:, 1:], g[f"{tag}/grad_stats"][:, 1:], rtol=2e-4, atol=1e-12)


def _run_oracle(cfg_over, A, B, n, gen_seed):
    cfg = dict(olosses.DEFAULT_CFG, **cfg_over)
    vit = _vit(cfg["dino_model_name"], cfg["dino_global_patch_size"])
    gp = {k: torch.from_numpy(v) for k, v in synth.generator_params(gen_seed, cfg["init_gain"]).items()}
    o = SpliceOracle(vit, gp, cfg)
    rows, grads = [], {}
    A, B = torch.from_numpy(A), torch.from_numpy(B)
    for _ in range(n):
        losses, _, g = o.step(A[None], B[None], A[None])
        rows.append(losses)
        grads[o.step_idx] = g
    with torch.no_grad():
        final = ogen.forward(o.params, A[None])
    return rows, grads, final


def _check_rows(rows, gl, keys, rtol):
    for i, r in enumerate(rows):
        for j, k in enumerate(keys):
            if np.isnan(gl[i, j]):
                assert k not in r, (i, k)
            else:
                np.testing.assert_allclose(r[k], gl[i, j], rtol=rtol, err_msg=f"step {i} {k}")


def test_steps_a_first_three(golden_dir):
    """Steps 0,1,2 of fixture (a): every entry of the reference's loss dict, and the
    generator gradients, agree; then the trajectory tracks for 12 steps."""
    g = _load(golden_dir, "steps.npz")
    keys = [str(k) for k in g["loss_keys"]]
    A, B = synth.smooth_image_pair(32, 0, 64, 64)
    rows, grads, _ = _run_oracle(dict(dino_model_name="dino_vits8", dino_global_patch_size=64), A, B, 12, 31)
    _check_rows(rows[:3], g["a/losses"], keys, 2e-4)
    for s in (0, 1, 2):
        got = np.stack([_stats(x) for x in grads[s]])
        ref = g[f"a/grad_stats_step{s}"]
        np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=5e-3, atol=1e-10)
    _check_rows(rows, g["a/losses"], keys, 2e-2)


def test_steps_b_resize_nonsquare(golden_dir):
    g = _load(golden_dir, "steps.npz")
    keys = [str(k) for k in g["loss_keys"]]
    A, B = synth.smooth_image_pair(34, 1, 48, 80)
    rows, grads, final = _run_oracle(dict(dino_model_name="dino_vits8", dino_global_patch_size=64), A, B, 4, 33)
    _check_rows(rows[:2], g["b/losses"], keys, 2e-4)
    _check_rows(rows, g["b/losses"], keys, 2e-2)
    assert final.shape == g["b/final_out"].shape


def test_general_skip_matches_reference_inversion_net(golden_dir):
    """splice_amd.unet_general.GeneralSkip (the non-default skip() used by inversion.py) against outputs and parameter
    gradients recorded from the reference's models/unet/skip.py with the same position-seeded parameters."""
    from splice_amd.networks import skip
    from oracle.fixtures import INVERSION_NET, sample, stats
    g = np.load(os.path.join(golden_dir, "inversion_net.npz"))
    with pytest.raises(RuntimeError, match="outside the HIP generator engine"):     # round 5: one backend -- outside the engine is an error by default
        skip(8, 3, device="cpu", **INVERSION_NET)
    os.environ["SPLICE_ALLOW_STOCK_TORCH"] = "1"
    try:
        with pytest.warns(RuntimeWarning, match="outside the HIP generator engine"):   # asked for by name, the stock-PyTorch net still announces itself
            net = skip(8, 3, device="cpu", **INVERSION_NET)
    finally:
        del os.environ["SPLICE_ALLOW_STOCK_TORCH"]
    params = list(net.named_parameters())
    assert len(params) == int(g["n_tensors"]) and sum(p.numel() for _, p in params) == int(g["n_params"])
    with torch.no_grad():
        for i, (name, p) in enumerate(params):
            off = 1.0 if p.dim() == 1 and name.endswith("weight") else 0.0
            p.copy_(torch.from_numpy(synth.normal(31, f"inv/p{i}", tuple(p.shape), 0.05, off)))
    for tag, (h, w) in {"96x72": (96, 72), "100x84": (100, 84)}.items():
        x = torch.from_numpy(synth.normal(32, "inv/x" + tag, (1, 8, h, w)))
        net.zero_grad()
        y = net(x)
        (y * y).mean().backward()
        assert y.shape == (1, 3, h, w)
        np.testing.assert_allclose(sample(y, 2053), g[f"{tag}/out_sample"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(stats(y), g[f"{tag}/out_stats"], rtol=1e-6)
        gs = np.stack([stats(p.grad) for _, p in params])
        np.testing.assert_allclose(gs[:, 1:], g[f"{tag}/grad_stats"][:, 1:], rtol=2e-4, atol=1e-10)   # |g| and g^2 sums


def test_define_g_seed_parity(golden_dir):
    """``torch.manual_seed(s); define_G(init_type, gain)`` gives the reference's initial generator (fixture recorded from
    models/networks.py:24-58 by oracle/make_golden.py): splice_amd.networks consumes the constructor draws of the reference's
    nn.Conv2d modules, then draws every tensor on the CPU generator in module order.  Bit-exact (same torch build)."""
    from splice_amd import networks
    g = np.load(os.path.join(golden_dir, "define_g_init.npz"))
    sample = lambda t, n=8: np.resize(t.reshape(-1)[::max(1, t.numel() // n)][:n].numpy(), n)
    for key in [k for k in g.files if k.endswith("/stats")]:
        init_type, seed, _ = key.split("/")
        torch.manual_seed(int(seed))
        from splice_amd.generator import DEFAULT_ARCH
        networks._constructor_state(DEFAULT_ARCH)      # what skip() draws while it builds the modules
        state = networks._draw_initial_state(init_type, 0.02)
        got_stats = np.stack([[t.double().sum().item(), t.double().abs().sum().item(), (t.double() ** 2).sum().item()] for t in state.values()])
        got_samples = np.stack([sample(t) for t in state.values()])
        assert np.array_equal(got_samples, g[f"{init_type}/{seed}/samples"]), key
        np.testing.assert_allclose(got_stats, g[key], rtol=1e-12, atol=0)


def test_skip_constructor_init_parity(golden_dir):
    """A fresh ``skip(...)`` (no init_weights: what inversion.py:21-25 trains from) carries the reference's constructor
    initialisation for a fixed seed -- kaiming-uniform conv weights, uniform biases, BatchNorm 1 / 0 -- and leaves the global
    generator where the reference leaves it.  Fixture recorded from models/unet/skip.py by oracle/make_golden.py.  (ADVICE r2:
    the HIP path used to return an all-zero arena for non-default architectures.)"""
    from oracle.fixtures import INVERSION_NET
    from splice_amd import networks
    from splice_amd.generator import DEFAULT_ARCH, arch_param_specs
    g = np.load(os.path.join(golden_dir, "skip_constructor_init.npz"))
    inv = dict(num_input_channels=32, num_output_channels=3, num_channels_down=INVERSION_NET["num_channels_down"], num_channels_up=INVERSION_NET["num_channels_up"],
               num_channels_skip=INVERSION_NET["num_channels_skip"], filter_size_down=INVERSION_NET["filter_size_down"], filter_size_up=INVERSION_NET["filter_size_up"],
               filter_skip_size=1, pad="reflection")
    sample = lambda t, n=8: np.resize(t.reshape(-1)[::max(1, t.numel() // n)][:n].numpy(), n)
    for key in [k for k in g.files if k.endswith("/stats")]:
        tag, seed, _ = key.split("/")
        arch = inv if tag == "inversion" else DEFAULT_ARCH
        torch.manual_seed(int(seed))
        state = networks._constructor_state(arch)
        nxt = torch.rand(4).numpy()
        ordered = [state[name] for name, _, _ in arch_param_specs(arch)]          # parameters() order
        assert len(ordered) == len(state)
        got_stats = np.stack([[t.double().sum().item(), t.double().abs().sum().item(), (t.double() ** 2).sum().item()] for t in ordered])
        assert np.array_equal(np.stack([sample(t) for t in ordered]), g[f"{tag}/{seed}/samples"]), key
        np.testing.assert_allclose(got_stats, g[key], rtol=1e-12, atol=0)
        assert np.array_equal(nxt, g[f"{tag}/{seed}/next_draw"]), key
        assert all(float(t.abs().sum()) > 0 for (name, _, kind), t in zip(arch_param_specs(arch), ordered) if kind in ("conv_w", "bn_w"))


@pytest.mark.parametrize("case,tol", [("up128", 2e-5), ("down900", 1e-4), ("cap64x150", 2e-5)])
def test_resize_against_independent_numpy_restatement(golden_dir, case, tol):
    """a non-identity ``Resize`` used to be pinned against itself (oracle/make_golden.py hands the reference the oracle's own
    ``resize_shorter_edge`` as its torchvision).  ``oracle/resize_np.py`` restates torchvision 0.10's tensor Resize (size rule + bilinear,
    align_corners=False, no antialias) in plain numpy / float64; its outputs (``tests/golden/resize_np.npz``, oracle/make_resize_golden.py)
    pin the oracle's torch path here and the HIP kernels in tests/test_ops_gpu.py.  Tolerances: float32 source coordinates resolve to
    6e-5 at a 900-pixel plane (measured: 5.2e-5 there, 1e-5 at 128)."""
    from oracle import resize_np
    from oracle.make_resize_golden import CASES, case_input
    shape, size = CASES[case]
    want = _load(golden_dir, "resize_np.npz")[case]
    x = case_input(case)
    assert np.array_equal(resize_np.resize_shorter_edge(x, size, 480), want)   # the fixture is what the committed restatement produces
    got = olosses.resize_shorter_edge(torch.from_numpy(x), size, 480).numpy()
    assert got.shape == want.shape == (1,) + resize_np.resize_output_size(shape[1], shape[2], size, 480)
    assert np.abs(got - want).max() < tol, np.abs(got - want).max()
