"""Drop-in for ``models/networks.py`` + ``models/unet/skip.py`` (``define_G``, ``init_weights``,
``init_net``, ``skip``) on the HIP generator engine.

``netG`` is an ``nn.Module`` whose 112 parameters carry the reference's ``state_dict`` names
(``1.0.1.0.weight`` ... ``9.0.bias``, the numbering of ``models/unet/common.py:6-9``) and are
VIEWS into one flat fp32 arena, so ``torch.optim`` / ``state_dict`` / ``load_state_dict`` work
unchanged while the engine reads and writes the arena directly.  The fused engine implements the reference's
default architecture (``skip()`` with its default arguments, which is all ``define_G`` ever builds) and every other
architecture its kernels cover (e.g. the 6-scale reflection-padded net of ``inversion.py``); what they do not cover returns a
stock-PyTorch module (``unet_general.GeneralSkip``) -- only on request (``SPLICE_ALLOW_STOCK_TORCH=1``), an error otherwise.  A fresh ``skip()`` carries PyTorch's constructor initialisation.
"""
import math

import torch
import torch.nn as nn

from .generator import GeneratorEngine


class _Node(nn.Module):
    """Container mirroring one level of the reference's dotted module path."""


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, *params):
        plan = net._acquire(x.shape[0], x.shape[2], x.shape[3], need_grad)
        y = plan.forward(net.flat, x.contiguous().float())
        if net.training:
            net._track_running_stats(plan)
        ctx.net, ctx.plan, ctx.need_grad = net, plan, need_grad
        if not need_grad:
            net._release(plan)
        return y

    @staticmethod
    def backward(ctx, dy):
        net, plan = ctx.net, ctx.plan
        if not ctx.need_grad:
            return (None, None, None) + (None,) * len(net._plist)
        g = plan.backward(net.flat, dy.contiguous().float())
        net._release(plan)
        grads = tuple(g[off:off + n].view(p.shape) for (off, n), p in zip(net._slices, net._plist))
        return (None, None, None) + grads


class SkipGenerator(nn.Module):
    """The reference generator (``skip()`` defaults) backed by ``splice_gen_*``."""

    def __init__(self, device="cuda", arch=None):
        super().__init__()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("SkipGenerator (HIP engine) needs a GPU device; there is no CPU fallback")
        self.engine = GeneratorEngine(device=dev, arch=arch)
        generator_param_specs = lambda: self.engine.param_specs
        self.flat = torch.zeros(self.engine.numel, device=dev)
        # BatchNorm buffers (nn.BatchNorm2d keeps them in state_dict; train mode moves them with momentum 0.1 and nothing
        # ever reads them): views of one flat arena the engine updates in ONE launch per forward
        self.running = torch.zeros(self.engine.buffer_numel, device=dev)
        self._batches = torch.zeros(len(self.engine.buffer_table) // 2, dtype=torch.long, device=dev)
        self._plist, self._slices, self._kinds = [], [], {}
        self._bn_names = []
        for name, shape, kind in generator_param_specs():
            off, n = self.engine.table[name]
            p = nn.Parameter(self.flat[off:off + n].view(shape))
            self._register(name, p)
            self._plist.append(p)
            self._slices.append((off, n))
            self._kinds[name] = kind
            if kind == "bn_w":   # BatchNorm buffers exist in the reference state_dict (never consumed: train mode)
                base = name[:-len(".weight")]
                off, cnt = self.engine.buffer_table[base + ".running_mean"]
                self._register_buffer(base + ".running_mean", self.running[off:off + cnt])
                off, cnt = self.engine.buffer_table[base + ".running_var"]
                self.running[off:off + cnt] = 1.0
                self._register_buffer(base + ".running_var", self.running[off:off + cnt])
                self._register_buffer(base + ".num_batches_tracked", self._batches[len(self._bn_names)])
                self._bn_names.append(base)
        self._free = {}

    def _walk(self, dotted):
        parts = dotted.split(".")
        node = self
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        return node, parts[-1]

    def _register(self, dotted, param):
        node, leaf = self._walk(dotted)
        node.register_parameter(leaf, param)

    def _register_buffer(self, dotted, buf):
        node, leaf = self._walk(dotted)
        node.register_buffer(leaf, buf)

    def _acquire(self, N, H, W, need_grad):
        pool = self._free.setdefault((N, H, W, need_grad), [])
        if pool:
            return pool.pop()
        from .generator import GeneratorPlan
        # a batch is ONE call of the reference's net: nn.BatchNorm2d statistics over all N images
        return GeneratorPlan(self.engine, N, H, W, need_grad, batch_stats=N > 1)

    def _release(self, plan):
        self._free.setdefault((plan.N, plan.H, plan.W, bool(plan.need_grad)), []).append(plan)

    def _track_running_stats(self, plan):
        import ctypes as C
        from . import _lib
        plans = (C.c_void_p * 1)(plan.handle)
        _lib.check(_lib.lib().splice_gen_running_stats_update(plans, 1, _lib.ptr(self.running), 0, 0.1, _lib.current_stream()), "running_stats_update")
        self._batches += 1

    def forward(self, x):
        """x ``[N,3,H,W]`` in [0,1]: ONE call of the net, as ``netG(input['A_global'])`` (models/model.py:15).  With N > 1
        (``n_crops`` > 1 crops stacked by data/transforms.py:27) the train-mode BatchNorm layers take their statistics over
        the whole batch, as the reference's ``nn.BatchNorm2d`` does (N <= 8)."""
        if not x.is_cuda:
            raise RuntimeError("SkipGenerator: input must be on the GPU")
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._plist)
        return _GenFn.apply(x, self, need_grad, *self._plist)


def skip(num_input_channels=3, num_output_channels=3, num_channels_down=[16, 32, 64, 128, 128],
         num_channels_up=[16, 32, 64, 128, 128], num_channels_skip=[4, 4, 4, 4, 4], filter_size_down=3, filter_size_up=3,
         filter_skip_size=1, need_sigmoid=True, need_tanh=False, need_bias=True, pad='zero', upsample_mode='bilinear',
         downsample_mode='stride', act_fun='LeakyReLU', need1x1_up=True, device="cuda"):
    """``models/unet/skip.py:4-11`` on the HIP generator engine: the default arguments (``define_G``) and every other
    architecture the kernels cover (up to 6 scales, 1/3/5/7 filters, zero or reflection padding, stride-2 down-sampling,
    bilinear up-sampling, LeakyReLU, sigmoid head) -- in particular the feature-inversion net of inversion.py:21-25."""
    n = len(num_channels_down)
    as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * n
    arch = dict(num_input_channels=num_input_channels, num_output_channels=num_output_channels, num_channels_down=list(num_channels_down),
                num_channels_up=list(num_channels_up), num_channels_skip=list(num_channels_skip), filter_size_down=as_list(filter_size_down),
                filter_size_up=as_list(filter_size_up), filter_skip_size=filter_skip_size, pad=pad)
    on_hip = (torch.device(device).type == "cuda" and n <= 6 and len(num_channels_up) == n and len(num_channels_skip) == n
              and need_sigmoid and need_bias and need1x1_up and pad in ('zero', 'reflection') and filter_skip_size == 1
              and all(m == 'bilinear' for m in as_list(upsample_mode)) and all(m == 'stride' for m in as_list(downsample_mode))
              and act_fun == 'LeakyReLU' and all(k in (1, 3, 5, 7) for k in arch["filter_size_down"] + arch["filter_size_up"])
              and all(0 < c <= 128 for c in arch["num_channels_down"] + arch["num_channels_up"] + arch["num_channels_skip"])
              and 0 < num_input_channels <= 128 and 0 < num_output_channels <= 16)
    if not on_hip:
        # what the kernels do not cover (other down-samplers / activations / no sigmoid, or a CPU device).  Round 5: this FAILS by default -- the
        # package has one backend; a net built from stock PyTorch modules (splice_amd.unet_general, the architecture restated for tests and for
        # experiments outside the engine) has to be asked for by name: SPLICE_ALLOW_STOCK_TORCH=1, and it still announces itself.
        import os
        import warnings
        msg = ("splice_amd.networks.skip: this architecture / device is outside the HIP generator engine "
               f"(device={device!r}, scales={n}, pad={pad!r}, act_fun={act_fun!r}, downsample_mode={downsample_mode!r}, "
               f"upsample_mode={upsample_mode!r}, need_sigmoid={need_sigmoid})")
        if os.environ.get("SPLICE_ALLOW_STOCK_TORCH") != "1" or os.environ.get("SPLICE_STRICT_HIP") == "1":
            raise RuntimeError(msg + "; set SPLICE_ALLOW_STOCK_TORCH=1 to build it from stock PyTorch modules instead (not the HIP engine)")
        msg += ": building stock PyTorch modules instead"
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
        from .unet_general import GeneralSkip
        return GeneralSkip(num_input_channels, num_output_channels, num_channels_down, num_channels_up, num_channels_skip,
                           filter_size_down, filter_size_up, filter_skip_size, need_sigmoid, need_tanh, need_bias, pad,
                           upsample_mode, downsample_mode, act_fun, need1x1_up).to(device)
    from .generator import DEFAULT_ARCH
    # The reference's skip() returns modules carrying PyTorch's CONSTRUCTOR initialisation (kaiming-uniform conv weights,
    # uniform biases, BatchNorm gamma 1 / beta 0); define_G overwrites it through init_weights, inversion.py:21-25 trains
    # from it as it is.  Same values here, drawn from the global CPU generator in the reference's construction order.
    net = SkipGenerator(device=device, arch=None if arch == DEFAULT_ARCH else arch)
    state = _constructor_state(arch)
    with torch.no_grad():
        for name, p in zip(net._kinds, net._plist):
            p.copy_(state[name])
    return net


def _constructor_state(arch):
    """{state_dict name: CPU tensor} of a freshly CONSTRUCTED ``skip()`` net.  The reference builds it out of ``nn.Conv2d`` /
    ``nn.BatchNorm2d`` modules on the CPU, scale by scale (models/unet/skip.py:46-97: skip conv, the two down convs, the two up
    convs of scale i, then scale i + 1, ..., the head last; models/unet/common.py:121); every ``nn.Conv2d`` constructor draws its
    default weights and bias from the global CPU generator.  Building the same modules in the same order gives the same
    tensors for a fixed ``torch.manual_seed`` -- and leaves the generator where the reference leaves it, so that
    ``torch.manual_seed(s); define_G(...)`` continues on the reference's random stream."""
    down, up, skipc = arch["num_channels_down"], arch["num_channels_up"], arch["num_channels_skip"]
    kd, ku, ks = arch["filter_size_down"], arch["filter_size_up"], arch["filter_skip_size"]
    cv = ".1" if arch["pad"] == "reflection" else ".0"
    n, state, prefix = len(down), {}, ""

    def conv(name, cin, cout, k):
        m = nn.Conv2d(cin, cout, k)
        state[name + cv + ".weight"], state[name + cv + ".bias"] = m.weight.detach().clone(), m.bias.detach().clone()

    def bn(name, c):
        state[name + ".weight"], state[name + ".bias"] = torch.ones(c), torch.zeros(c)

    for i in range(n):
        cin = arch["num_input_channels"] if i == 0 else down[i - 1]
        k = up[i + 1] if i + 1 < n else down[i]
        p = prefix
        conv(p + "1.0.1", cin, skipc[i], ks); bn(p + "1.0.2", skipc[i])
        conv(p + "1.1.1", cin, down[i], kd[i]); bn(p + "1.1.2", down[i])
        conv(p + "1.1.4", down[i], down[i], kd[i]); bn(p + "1.1.5", down[i])
        bn(p + "2", skipc[i] + k)
        conv(p + "3", skipc[i] + k, up[i], ku[i]); bn(p + "4", up[i])
        conv(p + "6", up[i], up[i], 1); bn(p + "7", up[i])
        prefix += "1.1.7."
    conv("9", up[0], arch["num_output_channels"], 1)
    return state


_CONV_INIT = {  # init_type -> in-place initialiser of a conv weight (models/networks.py:30-39)
    'normal': lambda w, gain: nn.init.normal_(w, 0.0, gain),
    'xavier': lambda w, gain: nn.init.xavier_normal_(w, gain=gain),
    'kaiming': lambda w, gain: nn.init.kaiming_normal_(w, a=0, mode='fan_in'),
    'orthogonal': lambda w, gain: nn.init.orthogonal_(w, gain=gain),
}


def _draw_initial_state(init_type, init_gain, specs=None):
    """{state_dict name: CPU tensor} drawn from the global CPU generator in the reference's module order."""
    if init_type not in _CONV_INIT:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    if specs is None:
        from .synth import generator_param_specs
        specs = generator_param_specs()
    state = {}
    for name, shape, kind in specs:
        host = torch.zeros(shape)
        if kind == "conv_w":
            _CONV_INIT[init_type](host, init_gain)
        elif kind == "bn_w":
            nn.init.normal_(host, 1.0, init_gain)
        state[name] = host
    return state


def init_weights(net, init_type='normal', init_gain=0.02, debug=False):
    """``models/networks.py:24-47``: conv weights by ``init_type``, conv bias 0, BN gamma ~ N(1, gain), beta 0.  Values are
    drawn on the CPU in module order (the reference initialises on the CPU and moves the net afterwards), then uploaded
    into the arena."""
    state = _draw_initial_state(init_type, init_gain, net.engine.param_specs)
    with torch.no_grad():
        for name, p in zip(net._kinds, net._plist):
            p.copy_(state[name])


def init_net(net, init_type='normal', init_gain=0.02, debug=False, initialize_weights=True):
    if initialize_weights:
        init_weights(net, init_type, init_gain=init_gain, debug=debug)
    return net


def define_G(init_type='normal', init_gain=0.02, initialize_weights=True, device="cuda"):
    """``models/networks.py:56-58``."""
    net = skip(device=device)
    return init_net(net, init_type, init_gain, initialize_weights=initialize_weights)
