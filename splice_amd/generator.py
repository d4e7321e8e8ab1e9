"""Host side of the generator engine (C ABI ``splice_gen_*``): flat parameter arena in the
reference's ``netG.parameters()`` order, per-shape plans, forward / backward / Adam.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib


DEFAULT_ARCH = dict(num_input_channels=3, num_output_channels=3, num_channels_down=[16, 32, 64, 128, 128], num_channels_up=[16, 32, 64, 128, 128],
                    num_channels_skip=[4, 4, 4, 4, 4], filter_size_down=[3] * 5, filter_size_up=[3] * 5, filter_skip_size=1, pad='zero')


def arch_param_specs(arch):
    """(name, shape, kind) of every parameter of a ``skip()`` net in ``parameters()`` order with the reference's state_dict
    names (models/unet/skip.py:46-99 + the ``add`` numbering of models/unet/common.py:6-9): kind in conv_w / conv_b / bn_w / bn_b."""
    down, up, skipc = arch["num_channels_down"], arch["num_channels_up"], arch["num_channels_skip"]
    kd, ku, ks = arch["filter_size_down"], arch["filter_size_up"], arch["filter_skip_size"]
    cv = ".1" if arch["pad"] == "reflection" else ".0"      # Conv2d sits behind a ReflectionPad2d inside its Sequential
    n, specs = len(down), []

    def conv(name, cout, cin, k):
        specs.append((name + ".weight", (cout, cin, k, k), "conv_w"))
        specs.append((name + ".bias", (cout,), "conv_b"))

    def bn(name, c):
        specs.append((name + ".weight", (c,), "bn_w"))
        specs.append((name + ".bias", (c,), "bn_b"))

    def scale(i, cin, p):
        conv(p + "1.0.1" + cv, skipc[i], cin, ks); bn(p + "1.0.2", skipc[i])
        conv(p + "1.1.1" + cv, down[i], cin, kd[i]); bn(p + "1.1.2", down[i])
        conv(p + "1.1.4" + cv, down[i], down[i], kd[i]); bn(p + "1.1.5", down[i])
        k = down[i]
        if i < n - 1:
            scale(i + 1, down[i], p + "1.1.7.")
            k = up[i + 1]
        bn(p + "2", skipc[i] + k)
        conv(p + "3" + cv, up[i], skipc[i] + k, ku[i]); bn(p + "4", up[i])
        conv(p + "6" + cv, up[i], up[i], 1); bn(p + "7", up[i])

    scale(0, arch["num_input_channels"], "")
    conv("9" + cv, arch["num_output_channels"], up[0], 1)
    return specs


class GeneratorEngine:
    def __init__(self, device="cuda", arch=None):
        """``arch`` = None: the default ``skip()`` of ``define_G``; else a dict of ``skip()`` keyword arguments (lists per
        scale for channels and filter sizes, ``pad`` 'zero' | 'reflection'), e.g. the feature-inversion net."""
        self.device = torch.device(device)
        self.arch = dict(DEFAULT_ARCH if arch is None else arch)
        h = C.c_void_p()
        if arch is None:
            _lib.check(_lib.lib().splice_gen_create(C.byref(h)), "gen_create")
        else:
            a = _lib.GenArch()
            n = len(self.arch["num_channels_down"])
            if n > 6:
                raise NotImplementedError("the HIP generator covers up to 6 scales")
            a.n_scales, a.in_channels, a.out_channels = n, self.arch["num_input_channels"], self.arch["num_output_channels"]
            for i in range(n):
                a.down[i], a.up[i], a.skip[i] = self.arch["num_channels_down"][i], self.arch["num_channels_up"][i], self.arch["num_channels_skip"][i]
                a.filter_down[i], a.filter_up[i] = self.arch["filter_size_down"][i], self.arch["filter_size_up"][i]
            a.filter_skip, a.reflect = self.arch["filter_skip_size"], int(self.arch["pad"] == "reflection")
            _lib.check(_lib.lib().splice_gen_create_arch(C.byref(a), C.byref(h)), "gen_create_arch")
        self.handle = h
        self.in_channels, self.out_channels = self.arch["num_input_channels"], self.arch["num_output_channels"]
        L = _lib.lib()
        self.numel = L.splice_gen_param_count(h)
        self.table = OrderedDict()
        for i in range(L.splice_gen_num_tensors(h)):
            name, off, n = C.c_char_p(), C.c_longlong(), C.c_longlong()
            _lib.check(L.splice_gen_tensor_info(h, i, C.byref(name), C.byref(off), C.byref(n)))
            self.table[name.value.decode()] = (off.value, n.value)
        self.param_specs = arch_param_specs(self.arch)
        if [(n_, int(torch.Size(sh).numel())) for n_, sh, _ in self.param_specs] != [(n_, cnt) for n_, (_, cnt) in self.table.items()]:
            raise RuntimeError("generator engine: the parameter table of the library and arch_param_specs disagree")
        # BatchNorm buffers of netG.state_dict(): "<bn>.running_mean" / "<bn>.running_var" -> (offset, numel) in the buffer arena
        self.buffer_numel = L.splice_gen_buffer_count(h)
        self.buffer_table = OrderedDict()
        for i in range(L.splice_gen_num_buffers(h)):
            name, off, n = C.c_char_p(), C.c_longlong(), C.c_longlong()
            _lib.check(L.splice_gen_buffer_info(h, i, C.byref(name), C.byref(off), C.byref(n)))
            self.buffer_table[name.value.decode()] = (off.value, n.value)
        self._plans = {}

    def __del__(self):
        try:
            self._plans.clear()
            if getattr(self, "handle", None):
                _lib.lib().splice_gen_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def flatten(self, state):
        """name -> array/tensor mapping (reference state_dict names) -> flat fp32 arena on the device."""
        flat = torch.zeros(self.numel, device=self.device)
        for name, (off, n) in self.table.items():
            t = torch.as_tensor(state[name]).to(self.device, torch.float32).reshape(-1)
            assert t.numel() == n, (name, t.numel(), n)
            flat[off:off + n] = t
        return flat

    def unflatten(self, flat, shapes=None):
        out = OrderedDict()
        for name, (off, n) in self.table.items():
            v = flat[off:off + n]
            out[name] = v.view(shapes[name]) if shapes else v
        return out

    def plan(self, N, H, W, need_grad=True):
        key = (N, H, W, bool(need_grad))
        if key not in self._plans:
            self._plans[key] = GeneratorPlan(self, N, H, W, need_grad)
        return self._plans[key]


class GeneratorPlan:
    def __init__(self, engine, N, H, W, need_grad, arena_stride=0, batch_stats=False):
        """``arena_stride`` > 0: the N images are INDEPENDENT generators -- image n uses ``params[n * stride:]`` and its
        gradient goes to ``grads[n * stride:]`` (several pairs in one launch); 0: one generator applied to N images."""
        self.engine, self.N, self.H, self.W, self.need_grad, self.arena_stride = engine, N, H, W, need_grad, arena_stride
        h = C.c_void_p()
        _lib.check(_lib.lib().splice_gen_plan_create(engine.handle, N, H, W, int(need_grad), C.byref(h)), "gen_plan_create")
        self.handle = h
        if arena_stride:
            _lib.check(_lib.lib().splice_gen_plan_set_arena_stride(h, arena_stride), "gen_plan_set_arena_stride")
        self.batch_stats = bool(batch_stats)
        if batch_stats:   # ONE netG call on the batch: BatchNorm statistics over all N images (the reference with n_crops > 1)
            _lib.check(_lib.lib().splice_gen_plan_set_batch_stats(h, 1), "gen_plan_set_batch_stats")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().splice_gen_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, params, x):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (self.N, self.engine.in_channels, self.H, self.W)
        y = torch.empty(self.N, self.engine.out_channels, self.H, self.W, device=x.device)
        _lib.check(_lib.lib().splice_gen_forward(self.handle, _lib.ptr(params), _lib.ptr(x), _lib.ptr(y), _lib.current_stream()), "gen_forward")
        return y

    def backward(self, params, dy, grads=None, accumulate=False):
        assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (self.N, self.engine.out_channels, self.H, self.W)
        if grads is None:
            grads = torch.zeros(self.N * self.arena_stride if self.arena_stride else self.engine.numel, device=self.engine.device)
        _lib.check(_lib.lib().splice_gen_backward(self.handle, _lib.ptr(params), _lib.ptr(dy), _lib.ptr(grads), int(accumulate),
                                                  _lib.current_stream()), "gen_backward")
        return grads


def adam_step(params, grads, m, v, lr, beta1, beta2, eps, step, zero_grad=False):
    _lib.check(_lib.lib().splice_adam_step(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), params.numel(), lr, beta1, beta2,
                                           eps, step, int(zero_grad), _lib.current_stream()), "adam_step")
