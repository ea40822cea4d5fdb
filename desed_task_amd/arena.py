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
        self.successor = None           # set by the owner when it replaces this arena (e.g. after a device move)
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
    With an arena whose gradients are flat the step is ONE launch; otherwise one launch per tensor.

    State: the moments live in two flat buffers shaped like the arena; `self.state[p]` holds, per parameter, views into
    them under torch.optim.Adam's own keys (`step`, `exp_avg`, `exp_avg_sq`), so `state_dict()` / `load_state_dict()` carry the
    full optimizer state in Adam's layout (what Lightning writes into a checkpoint's `optimizer_states` and reads back on
    resume, train_sed.py:293) and the per-tensor path continues from the same moments.  `arena` may be the ParamArena or
    the model that owns it (`model.arena` is then looked up at every step, so a `.to()` that rebuilt the arena is followed)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, arena=None, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self._arena_src = arena
        self.grad_scale = grad_scale
        self._flat_state = None

    # ---- a torch.optim.Adam handed in by the recipe (train_sed.py:199-201) --------------------------------------------
    served = False      # set by SEDTask4's whole-step mode: the update of this step() call has already run (inside the closure)

    @classmethod
    def adopt(cls, opt, model):
        """Turn `opt` -- the `torch.optim.Adam(sed_student.parameters(), lr, betas=(0.9, 0.999))` of recipes/dcase2023_task4_baseline/
        train_sed.py:199-201 -- into this class IN PLACE, so that every reference to it (the recipe's ExponentialWarmup, Lightning's
        wrapper, `configure_optimizers`) keeps pointing at the same object while step() becomes the one-launch arena update.
        `param_groups` (with torch's own keys: the state dict stays loadable by torch.optim.Adam) and any existing per-parameter
        state are kept.  -> True when `opt` now is (or already was) a FusedAdam; False -- `opt` untouched -- when it is not a plain
        Adam over exactly `model.parameters()` (weight decay, amsgrad, maximize, several groups, a tensor lr ...)."""
        if isinstance(opt, cls):
            if opt._arena_src is None:
                opt._arena_src = model
            return True
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1 or not hasattr(model, "arena"):
            return False
        g = opt.param_groups[0]
        if (g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable")
                or torch.is_tensor(g["lr"])):
            return False
        params = list(model.parameters())
        if len(g["params"]) != len(params) or any(a is not b for a, b in zip(g["params"], params)):
            return False
        opt.__class__ = cls
        opt._arena_src, opt.grad_scale, opt._flat_state = model, 1.0, None
        opt._patch_step_function()          # (what Optimizer.__init__ / __setstate__ do for a class: the profiler-hooked step)
        return True

    # ---- arena lookup ---------------------------------------------------------------------------------
    @property
    def arena(self):
        src = self._arena_src
        if src is None or isinstance(src, ParamArena):
            while isinstance(src, ParamArena) and src.successor is not None:     # the owner rebuilt it (device move)
                src = self._arena_src = src.successor
            return src
        return src.arena

    @arena.setter
    def arena(self, value):
        self._arena_src = value

    def _group_params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _flat_ok(self, arena):
        gp = self._group_params()
        return (arena is not None and len(self.param_groups) == 1 and len(gp) == len(arena.params)
                and all(a is b for a, b in zip(gp, arena.params)) and arena.is_intact())

    def _flat(self, arena):
        """The flat moment buffers for `arena` (created on first use; moved when the arena moved; seeded from per-tensor state
        when that is where the moments currently live, e.g. right after load_state_dict)."""
        st = self._flat_state
        if st is None:
            st = self._flat_state = dict(step=0, m=torch.zeros_like(arena.flat), v=torch.zeros_like(arena.flat))
            for p, o in zip(arena.params, arena.offsets):
                ps = self.state.get(p)
                if ps and "exp_avg" in ps:
                    st["m"][o:o + p.numel()].view(p.shape).copy_(ps["exp_avg"])
                    st["v"][o:o + p.numel()].view(p.shape).copy_(ps["exp_avg_sq"])
                    st["step"] = int(ps["step"])
            self._bind_views(arena)
        elif st["m"].device != arena.flat.device or st["m"].numel() != arena.flat.numel():
            if st["m"].numel() != arena.flat.numel():
                raise RuntimeError("FusedAdam: the parameter arena changed size")
            st["m"], st["v"] = st["m"].to(arena.flat.device), st["v"].to(arena.flat.device)
            self._bind_views(arena)
        return st

    def _bind_views(self, arena):
        st = self._flat_state
        for p, o in zip(arena.params, arena.offsets):
            self.state[p] = {"step": st["step"], "exp_avg": st["m"][o:o + p.numel()].view(p.shape),
                             "exp_avg_sq": st["v"][o:o + p.numel()].view(p.shape)}

    # ---- (de)serialisation in torch.optim.Adam's layout -------------------------------------------------------
    def state_dict(self):
        st = self._flat_state
        if st is not None:
            for ps in self.state.values():
                ps["step"] = torch.tensor(float(st["step"]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)         # fills self.state[p] with fresh tensors on each parameter's device
        steps = [int(ps["step"]) for ps in self.state.values() if "step" in ps]
        arena = self.arena
        if self._flat_state is not None and self._flat_ok(arena):
            # keep the flat buffers (a captured hipGraph holds their addresses): copy the loaded moments in place
            st = self._flat_state
            for p, o in zip(arena.params, arena.offsets):
                ps = self.state.get(p)
                if ps and "exp_avg" in ps:
                    st["m"][o:o + p.numel()].view(p.shape).copy_(ps["exp_avg"])
                    st["v"][o:o + p.numel()].view(p.shape).copy_(ps["exp_avg_sq"])
            st["step"] = steps[0] if steps else 0
            self._bind_views(arena)
        else:
            self._flat_state = None                 # rebuilt from self.state at the next flat step

    # ---- the step -------------------------------------------------------------------------------------------
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
        if self.served:
            # SEDTask4's whole-step mode: the closure's training_step ran the whole optimisation step -- this update included
            self.served = False
            return loss
        arena = self.arena
        if self._flat_ok(arena) and arena.grads_are_flat():
            st = self._flat(arena)
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
        # per-tensor path: continues from whatever moments exist (the flat buffers' views, a loaded state dict, or zeros)
        flat_step = self._flat_state["step"] if self._flat_state is not None else None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                step = int(flat_step if flat_step is not None else st["step"]) + 1
                st["step"] = step
                if st["exp_avg"].device != p.device:
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].to(p.device), st["exp_avg_sq"].to(p.device)
                g = p.grad.contiguous()
                self._launch(p.data, g, st["exp_avg"], st["exp_avg_sq"], p.numel(), group, step)
        if flat_step is not None:
            self._flat_state["step"] = flat_step + 1
        return loss
