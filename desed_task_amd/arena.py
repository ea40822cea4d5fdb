"""Flat parameter arena + fused optimizer / EMA (K10, K11).

All 62 parameter tensors of a CRNN are views into ONE contiguous fp32 buffer (16-byte aligned slots), and
so are their gradients.  That makes
  * the EMA teacher update (sed_trainer.py:187-199) one kernel instead of 124 tiny ops,
  * Adam (train_sed.py:199-201) one kernel,
  * the data-parallel gradient exchange one (or two bucketed) RCCL all-reduce over one tensor,
while every parameter keeps its reference name/shape, so state dicts and `torch.optim.*` still work.
"""
import math

import torch

from . import _lib
from . import graph as _graph


class ParamArena:
    def __init__(self, params):
        self.params = [p for p in params]
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self._index = {}
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self._index[p.data_ptr()] = o

    @property
    def device(self):
        return self.flat.device

    def is_intact(self):
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def grad_view_for(self, tensor):
        """View into flat_grad for the parameter whose storage `tensor` is -- only while that parameter has no
        .grad yet (otherwise autograd will accumulate and the kernels must not overwrite it)."""
        o = self._index.get(tensor.data_ptr())
        if o is None:
            return None
        i = self.offsets.index(o)
        p = self.params[i]
        if p.grad is not None or p.shape != tensor.shape:
            return None
        return self.flat_grad[o:o + p.numel()].view(p.shape)

    def grads_are_flat(self):
        base = self.flat_grad.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                return False
        return True

    def gather_grads(self):
        """Make flat_grad hold every parameter's gradient (no-op when the kernels already wrote it there)."""
        if self.grads_are_flat():
            return self.flat_grad
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                dst = self.flat_grad[o:o + p.numel()].view(p.shape)
                if p.grad is None:
                    dst.zero_()
                elif p.grad.data_ptr() != dst.data_ptr():
                    dst.copy_(p.grad)
        return self.flat_grad


def ema_update_(teacher_params, student_params, alpha, teacher_arena=None, student_arena=None):
    """theta_t <- alpha*theta_t + (1-alpha)*theta_s.  One launch when both models sit in matching arenas."""
    lib = _lib.get()
    a, oma = float(alpha), float(1.0 - alpha)
    a_dev = getattr(alpha, "dev", None)              # graph.DynFloat: the kernel reads alpha from device memory
    if (teacher_arena is not None and student_arena is not None and teacher_arena.numel == student_arena.numel
            and teacher_arena.offsets == student_arena.offsets and teacher_arena.is_intact() and student_arena.is_intact()):
        t, s = teacher_arena.flat, student_arena.flat
        lib.call("sed_ema_update", t.data_ptr(), s.data_ptr(), t.numel(), a, oma, a_dev, _lib.stream_ptr(t))
        return 1
    n = 0
    for pt, ps in zip(teacher_params, student_params):
        td, sd = pt.data, ps.data
        if not (td.is_contiguous() and sd.is_contiguous()):
            raise RuntimeError("ema_update_: parameters must be contiguous")
        _lib.check_tensor(td, "teacher parameter")
        if td.data_ptr() % 16 == 0 and sd.data_ptr() % 16 == 0:
            lib.call("sed_ema_update", td.data_ptr(), sd.data_ptr(), td.numel(), a, oma, a_dev, _lib.stream_ptr(td))
        else:   # unaligned tail-only form
            lib.call("sed_ema_update", td.data_ptr(), sd.data_ptr(), min(td.numel(), 3), a, oma, a_dev, _lib.stream_ptr(td))
            if td.numel() > 3:
                raise RuntimeError("ema_update_: unaligned parameter storage")
        n += 1
    return n


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps) semantics (no weight decay / amsgrad) on the HIP kernel.
    With an arena whose gradients are flat the step is ONE launch; otherwise one launch per tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, arena=None, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self.arena = arena
        self.grad_scale = grad_scale
        self._flat_state = None

    def _launch(self, p, g, m, v, n, group, step, hyper_dev=None):
        b1, b2 = group["betas"]
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        _lib.get().call("sed_adam_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, float(b1), float(b2),
                        float(group["eps"]), float(group["lr"] / bc1), float(1.0 / math.sqrt(bc2)), float(self.grad_scale),
                        hyper_dev, _lib.stream_ptr(p))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        arena = self.arena
        group_params = [p for g in self.param_groups for p in g["params"]]
        if (arena is not None and len(self.param_groups) == 1 and len(group_params) == len(arena.params)
                and all(a is b for a, b in zip(group_params, arena.params)) and arena.is_intact() and arena.grads_are_flat()):
            if self._flat_state is None:
                self._flat_state = dict(step=0, m=torch.zeros_like(arena.flat), v=torch.zeros_like(arena.flat))
            st = self._flat_state
            dyn = _graph.active()
            if dyn is not None:
                # hipGraph replay: step counter and the two step-dependent factors are host logic re-run every step
                group = self.param_groups[0]
                b1, b2 = group["betas"]

                def advance():
                    st["step"] += 1
                    dyn.hf[dyn.F_ADAM_STEP] = group["lr"] / (1.0 - b1 ** st["step"])
                    dyn.hf[dyn.F_ADAM_IBC2] = 1.0 / math.sqrt(1.0 - b2 ** st["step"])

                dyn.host(advance)
                self._launch(arena.flat, arena.flat_grad, st["m"], st["v"], arena.numel, group, st["step"],
                             hyper_dev=dyn.ptr(dyn.F_ADAM_STEP))
                return loss
            st["step"] += 1
            self._launch(arena.flat, arena.flat_grad, st["m"], st["v"], arena.numel, self.param_groups[0], st["step"])
            return loss
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0 if self._flat_state is None else self._flat_state["step"]
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad.contiguous()
                self._launch(p.data, g, st["m"], st["v"], p.numel(), group, st["step"])
        return loss
