"""Host side of the generator engine (C ABI ``splice_gen_*``): flat parameter arena in the
reference's ``netG.parameters()`` order, per-shape plans, forward / backward / Adam.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib


class GeneratorEngine:
    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        h = C.c_void_p()
        _lib.check(_lib.lib().splice_gen_create(C.byref(h)), "gen_create")
        self.handle = h
        L = _lib.lib()
        self.numel = L.splice_gen_param_count(h)
        self.table = OrderedDict()
        for i in range(L.splice_gen_num_tensors(h)):
            name, off, n = C.c_char_p(), C.c_longlong(), C.c_longlong()
            _lib.check(L.splice_gen_tensor_info(h, i, C.byref(name), C.byref(off), C.byref(n)))
            self.table[name.value.decode()] = (off.value, n.value)
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
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (self.N, 3, self.H, self.W)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().splice_gen_forward(self.handle, _lib.ptr(params), _lib.ptr(x), _lib.ptr(y), _lib.current_stream()), "gen_forward")
        return y

    def backward(self, params, dy, grads=None, accumulate=False):
        assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (self.N, 3, self.H, self.W)
        if grads is None:
            grads = torch.zeros(self.N * self.arena_stride if self.arena_stride else self.engine.numel, device=self.engine.device)
        _lib.check(_lib.lib().splice_gen_backward(self.handle, _lib.ptr(params), _lib.ptr(dy), _lib.ptr(grads), int(accumulate),
                                                  _lib.current_stream()), "gen_backward")
        return grads


def adam_step(params, grads, m, v, lr, beta1, beta2, eps, step, zero_grad=False):
    _lib.check(_lib.lib().splice_adam_step(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), params.numel(), lr, beta1, beta2,
                                           eps, step, int(zero_grad), _lib.current_stream()), "adam_step")
