"""torch.autograd bindings of the CNN / GRU / head kernels (C-ABI in include/sed_hip.h).

PyTorch here is plumbing: it owns the HBM buffers, the stream and the autograd tape; every FLOP of the
path runs in libsed_hip.so.  Activations are channels-last (B, T, F, C) fp32.
"""
import os

import torch

from . import _lib
from . import graph as _graph

BN_EPS = 1e-3        # desed_task/nnet/CNN.py:76
BN_MOMENTUM = 0.99
BLOCK0_FUSED = os.environ.get("SED_BLOCK0_FUSED", "1") != "0"     # first block without its pre-BN tensor in HBM (A/B switch)
# blocks 1-6, training mode, split-bf16 convolutions: the BatchNorm backward runs inside the data-gradient convolution's operand
# staging (sed_conv3x3_bf16x3_bnbwd) instead of as its own in-place pass over dz (A/B switch; bench.py --no-bn-fold)
BN_BWD_FOLD = os.environ.get("SED_BN_BWD_FOLD", "1") != "0"


GRU_DW_ATOMIC = False            # A/B switch (bench.py --gru-dw-atomic): the zero-fill + atomic split-K weight-gradient GEMMs

# Weight-gradient GEMMs of a BiGRU layer on a side HIP stream: they only feed the optimizer, while the next thing on the chain is
# the other layer's recurrence (96 workgroups: 160 CUs idle).  Only launcher.StepDriver.backward_joined() turns it on, for the
# duration of its loss.backward(), and joins the side stream before returning (same-box step 4.17 -> 4.13 ms).
GRU_DW_SIDE = False
GRU_DX_SPLITK = True             # bench.py --no-dx-splitk (A/B): the BiGRU dX product as one K slice
GRU_DW_SIDE_ALLOWED = True       # bench.py --no-gru-dw-side (A/B)
# Round 5: the weight-gradient GEMMs of CNN blocks 1-6 take the same way out (they were ~390 us of a backward chain that ran one kernel
# at a time).  Needs the side stream, i.e. only acts while GRU_DW_SIDE is on; bench.py --no-cnn-dw-side / SED_CNN_DW_SIDE=0 (A/B).
CNN_DW_SIDE = os.environ.get("SED_CNN_DW_SIDE", "1") != "0"
CNN_DW_SIDE_NOW = True           # set by the step driver around its backward(): off under a gradient exchange (the buckets of an overlapped
                                 # all-reduce are issued from inside the backward pass; nothing there waits for the side stream)
# (Measured and removed again: a side stream of their own -- a replayed graph ran the two lanes one after the other, 3.17 vs 3.04 ms --
#  and deferring only some of the six blocks: all six was best.  DESIGN.md 12.6.)
_side = {}



# Diagnostics hook (tools/ts_probe.py): a callable(tag) invoked on the stream that is current at a few points of the step -- the probe
# tool enqueues a one-thread kernel there that writes the GPU's wall clock, so that a hipGraph replay can be timed from the inside
# (HIP events cannot be placed in a replay, and rocprofv3 changes the very queue behaviour under study).  None in normal operation.
PROBE = None


def probe(tag):
    if PROBE is not None:
        PROBE(tag)


def side_stream(device):
    if device.type != "cuda":
        return None
    key = (device.type, device.index)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


def join_side_stream(device):
    """The current stream waits for the weight-gradient GEMMs launched on the side stream (no-op if there were none)."""
    s = _side.get((device.type, device.index))
    if _deferred:
        flush_deferred(s)
    if s is not None:
        torch.cuda.current_stream(device).wait_stream(s)


# Small reductions whose results only the optimizer reads (the head's weight gradients, the BiGRU bias gradients: 9 - 19 us each between
# two kernels of the backward chain).  While GRU_DW_SIDE is on they are parked here and launched on the side stream at the next point
# where that stream has already been forked from the chain -- the weight-gradient section of the next BiGRU layer backward, or
# join_side_stream() at the latest.  (A fork of their own would cost what they save: a replayed graph keeps the first-captured
# successor of a fork on the main hardware queue and starts the other late -- DESIGN.md, "A hipGraph finding".)
# The same idea for the FORWARD half of a step: launches whose results nothing on the chain reads (the eight loss sums) are parked here
# while a step driver that has a side stream for the EMA runs training_step(), and go out on that stream with the EMA.  None: nobody
# collects them -- launch in place.
AFTER_FORWARD = None
PARK_LOSS_SUMS = True            # bench.py --no-park-loss (A/B)


def run_after_forward(parked, side):
    """Launch what a training_step() parked in AFTER_FORWARD on `side` (the caller has made it wait for the main stream); side = None
    (SIDE_ON_CPU test hook): in place."""
    for launch, keep in parked:
        if side is None:
            launch(_lib.stream_ptr(keep[0]))
            continue
        for t in keep:
            t.record_stream(side)
        with torch.cuda.stream(side):
            launch(side.cuda_stream)


_deferred = []
DEFER_OFF_CHAIN = True           # bench.py --no-defer (A/B): launch them where they are produced, on the chain


SIDE_ON_CPU = False              # TEST HOOK (tests/test_emu_step.py): run the PARKING logic of the side-stream launches on the CPU emulator,
                                 # where there are no streams -- parked launches then go out at the flush points, in place


def defer_off_chain(device, launch, keep):
    """launch(stream_ptr) now, or -- while the side stream is in use (GRU_DW_SIDE) -- later on that stream.  `keep`: the scratch tensors
    the launch reads (held until then, and marked as used by the side stream).  Gradient OUTPUTS must be passed to `launch` as
    addresses: they are returned to autograd, which only adopts a tensor nobody else references."""
    if GRU_DW_SIDE and DEFER_OFF_CHAIN and (device.type == "cuda" or SIDE_ON_CPU):
        fork = None
        if device.type == "cuda":
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(device))     # the side stream forks HERE, whenever the launch is enqueued
        _deferred.append((launch, keep, fork))
    else:
        launch(_lib.stream_ptr(keep[0]))


def flush_deferred(side):
    """Launch what defer_off_chain() parked: on `side`, each behind the point of the chain where it was parked -- or on the current
    stream (side = None)."""
    todo, _deferred[:] = list(_deferred), []
    for launch, keep, fork in todo:
        if side is not None:
            side.wait_event(fork)
            for t in keep:
                t.record_stream(side)
            with torch.cuda.stream(side):
                launch(side.cuda_stream)
        else:
            launch(_lib.stream_ptr(keep[0]))


def gemm_entry(cfg, pair=True):
    """C-ABI entry of the GRU GEMMs: split-bf16 products by default (fp32-level accuracy, see sed_gemm_bf16.hip),
    exact-f32 MFMA with cfg["gemm_precision"] = "f32" or SED_GEMM_PRECISION=f32."""
    prec = (cfg or {}).get("gemm_precision") or os.environ.get("SED_GEMM_PRECISION", "bf16x3")
    if prec not in ("bf16x3", "f32"):
        raise ValueError("gemm_precision must be 'bf16x3' or 'f32'")
    name = "sed_gemm_pair" if pair else "sed_gemm"
    return name + "_bf16x3" if prec == "bf16x3" else name


def _p(t):
    return None if t is None else t.data_ptr()



def dropout_params(p):
    """-> (thr24, dscale): keep element e iff (hash(e, seed) >> 8) >= thr24."""
    if p <= 0.0:
        return 0, 1.0
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return int(round(p * (1 << 24))), 1.0 / (1.0 - p)


_seed_gens = {}                 # stream name -> (generator, (torch seed, RANK) it was derived from)
_seed_stream = "default"


def reseed_dropout():
    """Restart the private dropout-seed streams from the current (torch seed, RANK).  seed_generator() does this on its own when
    torch.manual_seed() was called with a DIFFERENT seed; a second run under the SAME seed inside one process must call this (or
    restore `dropout_rng_state`) to see the same dropout masks again."""
    _seed_gens.clear()
    return seed_generator()


def dropout_rng_state():
    """State of the dropout-seed streams ({name: CPU generator state tensor}) -- saved in launcher.checkpoint_dict."""
    seed_generator()
    return {name: g.get_state().clone() for name, (g, _) in _seed_gens.items()}


def set_dropout_rng_state(state):
    if torch.is_tensor(state):                          # (checkpoints written before the streams were named)
        state = {"default": state}
    for name, st in state.items():
        seed_generator(name).set_state(st.clone().to(torch.uint8).cpu())


class seed_stream:
    """Context: the dropout / SpecAugment seeds drawn inside come from the private stream `name`.  The teacher's CNN forward draws
    from its own stream ("teacher_cnn"), so that WHEN it runs relative to the student's call sites -- inside the step, or one step
    ahead under the previous step's backward (SEDTask4's pipelined front-end) -- does not change any mask."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _seed_stream
        self.prev, _seed_stream = _seed_stream, self.name
        return self

    def __exit__(self, *exc):
        global _seed_stream
        _seed_stream = self.prev
        return False


def seed_generator(name=None):
    """The private CPU generator the dropout seeds of stream `name` (default: the current seed_stream) are drawn from.  It is
    derived from torch's seed (re-derived whenever torch.manual_seed() was called with another seed since; same seed again:
    reseed_dropout()), from RANK and from the stream's name, but it CONSUMES nothing from the global generator: the global CPU
    stream then sees exactly what the reference's step draws from it (the mixup permutations), and ranks seeded alike stay in
    lock-step whatever their dropout call sites do."""
    name = _seed_stream if name is None else name
    src = (torch.initial_seed(), os.environ.get("RANK", "0"))
    ent = _seed_gens.get(name)
    if ent is None or ent[1] != src:
        g = torch.Generator()
        salt = 0 if name == "default" else (int.from_bytes(name.encode()[:7], "little") * 2654435761)
        g.manual_seed((src[0] * 6364136223846793005 + 1442695040888963407 + 7919 * int(src[1] or 0) + salt) % (2 ** 63))
        ent = _seed_gens[name] = (g, src)
    return ent[0]


def new_seed(generator=None):
    """A fresh 31-bit dropout seed (host side, no device sync) from `generator` or the current private seed stream.
    Under a graph.DynArgs step the draw is repeated every replay (from the stream that was current at the call site) and the seed
    travels through device memory."""
    stream = _seed_stream

    def draw():
        g = generator if generator is not None else seed_generator(stream)
        return int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g).item())
    dyn = _graph.active()
    if dyn is not None:
        return dyn.new_seed(draw)
    return draw()


def _arena_views(cfg, *grads):
    """Are all these gradient buffers views into the parameter arena's flat gradient tensor?  Only then may their producer be launched
    LATER, off the chain, through raw addresses: autograd adopts such a view as .grad, so the address stays valid and is
    what the optimizer reads.  A buffer that is a fresh tensor instead -- the parameter already had a .grad (a second backward without
    zero_grad: AccumulateGrad adds it at once, on the main stream) or is frozen (the tensor is dropped after backward) -- must be
    written before this node returns."""
    arena = cfg.get("arena") if cfg else None
    if arena is None:
        return False
    lo = arena.flat_grad.data_ptr()
    hi = lo + 4 * arena.flat_grad.numel()
    return all(lo <= g.data_ptr() < hi for g in grads)


def _grad_buf(cfg, param):
    """Gradient output buffer for `param`: a view into the parameter arena's flat gradient buffer when the
    parameter lives in one (so the optimizer / all-reduce see ONE contiguous tensor), else a fresh tensor."""
    arena = cfg.get("arena") if cfg else None
    if arena is not None:
        g = arena.grad_view_for(param)
        if g is not None:
            return g
    return torch.empty_like(param)


def pack_conv_weights(weights, need_dgrad=True, precision="f32", prologue=None):
    """Repack a list of nn.Conv2d weights (COUT,CIN,3,3) into the conv kernels' layouts, ALL layers in ONE launch.
    precision "f32": K-major fp32 (sed_conv3x3); "bf16x3": split-bf16 slabs (sed_conv3x3_bf16x3).
    prologue (bf16x3 only): {"bounds": seeded SpecAugment draw, "copy": (src, dst)} done by the SAME launch (sed_cnn_prologue_bf16).
    -> [(Wf, Wd or None), ...] (views into one buffer; Wd = flipped/transposed pack for the data gradient)."""
    import ctypes
    n = len(weights)
    if n == 0:
        return []
    dev = weights[0].device
    sizes = [w.numel() for w in weights]
    buf = torch.empty((2 if need_dgrad else 1) * sum(sizes), device=dev, dtype=torch.float32)
    _lib.check_tensor(buf, "packed conv weights")
    W = (ctypes.c_void_p * n)(); Wf = (ctypes.c_void_p * n)(); Wd = (ctypes.c_void_p * n)()
    co = (ctypes.c_int * n)(); ci = (ctypes.c_int * n)()
    out, off = [], 0
    for k, w in enumerate(weights):
        w = w.contiguous()
        wf = buf[off:off + sizes[k]]; off += sizes[k]
        wd = None
        if need_dgrad:
            wd = buf[off:off + sizes[k]]; off += sizes[k]
        out.append((wf, wd))
        W[k] = w.data_ptr(); Wf[k] = wf.data_ptr(); Wd[k] = wd.data_ptr() if wd is not None else None
        co[k] = w.shape[0]; ci[k] = w.shape[1]
    if prologue is not None:
        if precision != "bf16x3":
            raise ValueError("the fused CNN prologue is built for the split-bf16 packs")
        b = prologue.get("bounds")                         # dict(out, n, f_param, n_freq, t_param, n_time, seed) or None
        c = prologue.get("copy")                           # (src, dst) or None
        seed = b["seed"] if b else 0
        _lib.get().call("sed_cnn_prologue_bf16", n, W, Wf, Wd, co, ci,
                        b["out"].data_ptr() if b else None, b["out"].shape[0] if b else 0, b["n"] if b else 1,
                        b["f_param"] if b else 0, b["n_freq"] if b else 1, b["t_param"] if b else 0, b["n_time"] if b else 1,
                        int(seed) & 0xFFFFFFFF, getattr(seed, "dev", None),
                        c[0].data_ptr() if c else None, c[1].data_ptr() if c else None, c[0].numel() if c else 0,
                        _lib.stream_ptr(buf))
        return out
    entry = "sed_conv_pack_multi_bf16" if precision == "bf16x3" else "sed_conv_pack_multi"
    _lib.get().call(entry, n, W, Wf, Wd, co, ci, _lib.stream_ptr(buf))
    return out


class ConvBlockFn(torch.autograd.Function):
    """One CNN block: Conv2d(3x3,p1) -> BatchNorm2d -> GLU -> Dropout -> AvgPool2d  (CNN.py:66-98).

    x: (B,T,F) for the first block (n_in_channel = 1; optional SpecAugment bounds fused into the load)
       or (B,T,F,CIN).  Returns (B, T//PT, F//PF, COUT).  Saves only x, the pre-BN conv output y and the
       batch statistics; the backward recomputes BN/GLU/sigmoid/dropout-mask on the fly."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, bn_w, bn_b, glu_w, glu_b, running_mean, running_var, cfg):
        lib = _lib.get()
        _lib.check_tensor(x, "block input")
        first = x.dim() == 3
        B, T, F = x.shape[0], x.shape[1], x.shape[2]
        COUT, CIN = conv_w.shape[0], conv_w.shape[1]
        PT, PF = cfg["pool"]
        training = bool(cfg["bn_training"])
        thr24, dscale = dropout_params(cfg.get("dropout_p", 0.0) if cfg.get("apply_dropout", False) else 0.0)
        seed = cfg.get("seed", 0)
        seed = seed if isinstance(seed, _graph.DynSeed) else int(seed)
        bounds = cfg.get("bounds") if first else None
        st = _lib.stream_ptr(x)
        dev = x.device
        # First block, fused: its pre-BN tensor y (246 MB at B = 48) never exists in HBM -- statistics pass, then one kernel for
        # conv + BN + GLU + dropout + pooling; the backward recomputes y from x (csrc/sed_block0.hip).  The unfused kernels stay
        # for what the fused ones do not cover (other widths / poolings, gradients through eval-mode BatchNorm).
        need_grad = any(ctx.needs_input_grad)
        if (first and COUT == 16 and (PT, PF) == (2, 2) and F % 8 == 0 and 8 <= F <= 128 and cfg.get("block0_fused", BLOCK0_FUSED)
                and (training or not need_grad)):
            conv_w = conv_w.contiguous()
            stats = torch.empty(4 * COUT, device=dev, dtype=torch.float32)
            nblk, partial = 0, None
            if training:
                nblk = lib.value("sed_conv_fwd_blocks", B, T, F, CIN, COUT)
                partial = torch.empty(nblk * 2 * COUT, device=dev, dtype=torch.float32)
                lib.call("sed_conv0_fwd", x.data_ptr(), conv_w.data_ptr(), _p(conv_b), _p(bounds), None, partial.data_ptr(),
                         B, T, F, COUT, st)
            lib.call("sed_bn_finalize", _p(partial), nblk, COUT, float(B * T * F), bn_w.data_ptr(), bn_b.data_ptr(),
                     running_mean.data_ptr(), running_var.data_ptr(), BN_MOMENTUM, BN_EPS, stats.data_ptr(), int(training),
                     int(training and cfg.get("update_running", True)), st)
            out = torch.empty(B, T // PT, F // PF, COUT, device=dev, dtype=torch.float32)
            glu_w = glu_w.contiguous()
            lib.call("sed_block0_fwd", x.data_ptr(), conv_w.data_ptr(), _p(conv_b), _p(bounds), stats.data_ptr(), glu_w.data_ptr(),
                     glu_b.data_ptr(), out.data_ptr(), B, T, F, int(seed), thr24, dscale, _graph.seed_dev(seed), st)
            ctx.save_for_backward(x, stats, conv_w, bn_w, bn_b, glu_w, glu_b, conv_b)
            ctx.meta = (first, B, T, F, CIN, COUT, PT, PF, training, seed, thr24, dscale, bounds)
            ctx.cfg = cfg
            ctx.fused0 = True
            return out
        ctx.fused0 = False
        y = torch.empty(B, T, F, COUT, device=dev, dtype=torch.float32)
        bf16x3 = (not first) and cfg.get("conv_precision", "f32") == "bf16x3" and cfg.get("packed") is not None
        nblk = lib.value("sed_conv_fwd_blocks_bf16" if bf16x3 else "sed_conv_fwd_blocks", B, T, F, CIN, COUT)
        partial = torch.empty(nblk * 2 * COUT, device=dev, dtype=torch.float32) if training else None
        conv_w = conv_w.contiguous()
        if first:
            lib.call("sed_conv0_fwd", x.data_ptr(), conv_w.data_ptr(), _p(conv_b), _p(bounds), y.data_ptr(), _p(partial),
                     B, T, F, COUT, st)
        else:
            packed = cfg.get("packed")
            if packed is not None:
                wf = packed[0]
            else:
                wf = torch.empty(9 * CIN * COUT, device=dev, dtype=torch.float32)
                lib.call("sed_conv_pack_weights", conv_w.data_ptr(), wf.data_ptr(), None, COUT, CIN, st)
            lib.call("sed_conv3x3_bf16x3" if bf16x3 else "sed_conv3x3", x.data_ptr(), wf.data_ptr(), _p(conv_b), y.data_ptr(), _p(partial), B, T, F, CIN, COUT, st)
        stats = torch.empty(4 * COUT, device=dev, dtype=torch.float32)
        lib.call("sed_bn_finalize", _p(partial), nblk, COUT, float(B * T * F), bn_w.data_ptr(), bn_b.data_ptr(),
                 running_mean.data_ptr(), running_var.data_ptr(), BN_MOMENTUM, BN_EPS, stats.data_ptr(), int(training),
                 int(training and cfg.get("update_running", True)), st)
        out = torch.empty(B, T // PT, F // PF, COUT, device=dev, dtype=torch.float32)
        glu_w = glu_w.contiguous()
        lib.call("sed_glu_fwd", y.data_ptr(), stats.data_ptr(), glu_w.data_ptr(), glu_b.data_ptr(), out.data_ptr(), B, T, F, COUT,
                 PT, PF, int(seed), thr24, dscale, _graph.seed_dev(seed), int(cfg.get("conv_precision", "f32") == "bf16x3"), st)
        ctx.save_for_backward(x, y, stats, conv_w, bn_w, bn_b, glu_w, glu_b, conv_b)
        ctx.meta = (first, B, T, F, CIN, COUT, PT, PF, training, seed, thr24, dscale, bounds)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.get()
        probe("convblock_bwd")
        if ctx.fused0:
            x, stats, conv_w, bn_w, bn_b, glu_w, glu_b, conv_b = ctx.saved_tensors
            first, B, T, F, CIN, COUT, PT, PF, training, seed, thr24, dscale, bounds = ctx.meta
            cfg = ctx.cfg
            gout = gout.contiguous()
            st = _lib.stream_ptr(x)
            d_w, d_bias = _grad_buf(cfg, conv_w), _grad_buf(cfg, conv_b)
            d_gamma, d_beta = _grad_buf(cfg, bn_w), _grad_buf(cfg, bn_b)
            d_glu_w, d_glu_b = _grad_buf(cfg, glu_w), _grad_buf(cfg, glu_b)
            scratch = torch.empty(int(lib.value("sed_block0_bwd_scratch_floats", B, T, F)), device=x.device, dtype=torch.float32)
            lib.call("sed_block0_bwd", x.data_ptr(), conv_w.data_ptr(), _p(conv_b), _p(bounds), stats.data_ptr(), bn_w.data_ptr(),
                     bn_b.data_ptr(), glu_w.data_ptr(), glu_b.data_ptr(), gout.data_ptr(), d_w.data_ptr(), d_bias.data_ptr(),
                     d_gamma.data_ptr(), d_beta.data_ptr(), d_glu_w.data_ptr(), d_glu_b.data_ptr(), scratch.data_ptr(), B, T, F,
                     int(seed), thr24, dscale, _graph.seed_dev(seed), st)
            if _deferred:           # block 1's weight gradient: beside this kernel
                flush_deferred(side_stream(x.device) if GRU_DW_SIDE else None)
            return None, d_w, d_bias, d_gamma, d_beta, d_glu_w, d_glu_b, None, None, None
        x, y, stats, conv_w, bn_w, bn_b, glu_w, glu_b, conv_b = ctx.saved_tensors
        first, B, T, F, CIN, COUT, PT, PF, training, seed, thr24, dscale, bounds = ctx.meta
        gout = gout.contiguous()
        dev = y.device
        st = _lib.stream_ptr(y)
        f32 = dict(device=dev, dtype=torch.float32)
        dz = torch.empty_like(y)
        cfg = ctx.cfg
        d_glu_w = _grad_buf(cfg, glu_w)
        d_glu_b = _grad_buf(cfg, glu_b)
        d_gamma = _grad_buf(cfg, bn_w)
        d_beta = _grad_buf(cfg, bn_b)
        nscr = int(lib.value("sed_glu_bwd_scratch_floats", B, T, F, COUT, PT, PF))
        gscratch = torch.empty(nscr, **f32) if nscr else None
        lib.call("sed_glu_bwd", y.data_ptr(), stats.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(), glu_w.data_ptr(), glu_b.data_ptr(),
                 gout.data_ptr(), dz.data_ptr(), d_glu_w.data_ptr(), d_glu_b.data_ptr(), d_gamma.data_ptr(), d_beta.data_ptr(),
                 _p(gscratch), B, T, F, COUT, PT, PF, int(seed), thr24, dscale, _graph.seed_dev(seed),
                 int(cfg.get("conv_precision", "f32") == "bf16x3"), st)
        if _deferred:           # the last BiGRU layer's side section: enqueued now that the chain's next kernel is
            flush_deferred(side_stream(dev) if GRU_DW_SIDE else None)
        d_bias = _grad_buf(cfg, conv_b)
        d_w = _grad_buf(cfg, conv_w)
        dx = None
        if first and training:
            # nobody but the weight gradient consumes dy of the first block: BN backward fused into its load
            lib.call("sed_conv0_wgrad", x.data_ptr(), _p(bounds), dz.data_ptr(), y.data_ptr(), stats.data_ptr(), bn_w.data_ptr(),
                     d_gamma.data_ptr(), d_beta.data_ptr(), d_w.data_ptr(), d_bias.data_ptr(), B, T, F, COUT, 1, 1, st)
            return dx, d_w, d_bias, d_gamma, d_beta, d_glu_w, d_glu_b, None, None, None
        packed = cfg.get("packed")
        if (BN_BWD_FOLD and training and not first and ctx.needs_input_grad[0] and cfg.get("conv_precision", "f32") == "bf16x3"
                and packed is not None and packed[1] is not None):
            # data gradient first: it forms dy = BN-backward(dz) while staging its operand and leaves dy behind for the weight
            # gradient (bit-identical to the separate sed_bn_bwd_apply pass: same expression, operation for operation)
            dy = torch.empty_like(dz)
            dx = torch.empty_like(x)
            lib.call("sed_conv3x3_bf16x3_bnbwd", dz.data_ptr(), y.data_ptr(), stats.data_ptr(), bn_w.data_ptr(), d_gamma.data_ptr(),
                     d_beta.data_ptr(), packed[1].data_ptr(), dx.data_ptr(), dy.data_ptr(), d_bias.data_ptr(), B, T, F, COUT, CIN, st)
            scratch = torch.empty(int(lib.value("sed_conv_wgrad_scratch_floats", B, T, F, CIN, COUT)), **f32)
            d_w_ptr = d_w.data_ptr()            # (an address, not the tensor, goes into what outlives this call: see defer_off_chain)

            def wgrad(stream_ptr):
                lib.call("sed_conv_wgrad_bf16x3", x.data_ptr(), dy.data_ptr(), scratch.data_ptr(), d_w_ptr, B, T, F, CIN, COUT, stream_ptr)
            if CNN_DW_SIDE and CNN_DW_SIDE_NOW and _arena_views(cfg, d_w):
                # the chain goes on with the block below (its GLU backward reads dx); dW only feeds the optimizer.  Parked until that
                # block's first kernel is enqueued, then launched on the side stream beside it (block 1's goes out beside block 0's
                # backward, which had the end of the step to itself)
                defer_off_chain(dev, wgrad, (x, dy, scratch))
            else:
                wgrad(st)
            return dx, d_w, d_bias, d_gamma, d_beta, d_glu_w, d_glu_b, None, None, None
        lib.call("sed_bn_bwd_apply", y.data_ptr(), dz.data_ptr(), stats.data_ptr(), bn_w.data_ptr(), d_gamma.data_ptr(),
                 d_beta.data_ptr(), d_bias.data_ptr(), B * T * F, COUT, int(training), st)
        dy = dz
        if first:
            lib.call("sed_conv0_wgrad", x.data_ptr(), _p(bounds), dy.data_ptr(), None, None, None, None, None, d_w.data_ptr(), None,
                     B, T, F, COUT, 0, int(training), st)
        else:
            scratch = torch.empty(int(lib.value("sed_conv_wgrad_scratch_floats", B, T, F, CIN, COUT)), **f32)
            entry = "sed_conv_wgrad_bf16x3" if cfg.get("conv_precision", "f32") == "bf16x3" else "sed_conv_wgrad"
            lib.call(entry, x.data_ptr(), dy.data_ptr(), scratch.data_ptr(), d_w.data_ptr(), B, T, F, CIN, COUT, st)
            if ctx.needs_input_grad[0]:
                packed = cfg.get("packed")
                bf16x3 = False
                if packed is not None and packed[1] is not None:
                    wd = packed[1]                # packed with the forward weights (same values: no optimizer step in between)
                    bf16x3 = cfg.get("conv_precision", "f32") == "bf16x3"
                else:
                    wd = torch.empty(9 * CIN * COUT, **f32)
                    wf = torch.empty(9 * CIN * COUT, **f32)
                    lib.call("sed_conv_pack_weights", conv_w.data_ptr(), wf.data_ptr(), wd.data_ptr(), COUT, CIN, st)
                dx = torch.empty_like(x)
                lib.call("sed_conv3x3_bf16x3" if bf16x3 else "sed_conv3x3", dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, B, T, F, COUT, CIN, st)
        return dx, d_w, d_bias, d_gamma, d_beta, d_glu_w, d_glu_b, None, None, None


def _gemm(lib, A, Bm, bias, C, M, N, K, lda, ldb, ldc, ta, tb, split_k, accumulate, st):
    lib.call("sed_gemm", A, Bm, bias, C, M, N, K, lda, ldb, ldc, ta, tb, split_k, accumulate, st)


class BiGRULayerFn(torch.autograd.Function):
    """One bidirectional GRU layer (nn.GRU semantics, RNN.py:19-30).  x (B,T,I) -> (B,T,2H), H = 128.
    Input projections and all weight gradients are MFMA GEMMs; the recurrence is the persistent kernel."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, cfg=None):
        lib = _lib.get()
        x = x.contiguous()
        _lib.check_tensor(x, "gru input")
        B, T, I = x.shape
        H = w_hh_f.shape[1]
        st = _lib.stream_ptr(x)
        f32 = dict(device=x.device, dtype=torch.float32)
        gi = torch.empty(B, T, 2, 3 * H, **f32)
        w_ih = [w_ih_f.contiguous(), w_ih_r.contiguous()]
        w_hh = [w_hh_f.contiguous(), w_hh_r.contiguous()]
        # both directions' input projections in one launch: gi[:, :, d, :] = x . W_ih[d]^T + b_ih[d]
        lib.call(gemm_entry(cfg), x.data_ptr(), x.data_ptr(), w_ih[0].data_ptr(), w_ih[1].data_ptr(), b_ih_f.data_ptr(),
                 b_ih_r.data_ptr(), gi.data_ptr(), gi.data_ptr() + 3 * H * 4, B * T, 3 * H, I, I, I, 6 * H, 0, 1, 1, 0, st)
        out = torch.empty(B, T, 2 * H, **f32)
        need = any(ctx.needs_input_grad)
        saved = torch.empty(B, T, 2, 4, H, **f32) if need else None
        lib.call("sed_gru_fwd", gi.data_ptr(), w_hh[0].data_ptr(), w_hh[1].data_ptr(), b_hh_f.data_ptr(), b_hh_r.data_ptr(),
                 out.data_ptr(), _p(saved), B, T, H, st)
        if need:
            ctx.save_for_backward(x, out, saved, w_ih[0], w_ih[1], w_hh[0], w_hh[1], b_ih_f, b_hh_f, b_ih_r, b_hh_r)
        ctx.dims = (B, T, I, H)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.get()
        probe("bigru_bwd")
        x, out, saved, w_ih_f, w_ih_r, w_hh_f, w_hh_r, b_ih_f, b_hh_f, b_ih_r, b_hh_r = ctx.saved_tensors
        B, T, I, H = ctx.dims
        cfg = ctx.cfg
        dout = dout.contiguous()
        st = _lib.stream_ptr(x)
        f32 = dict(device=x.device, dtype=torch.float32)
        dgi = torch.empty(B, T, 2, 3 * H, **f32)
        dgh = torch.empty(B, T, 2, 3 * H, **f32)
        hprev = torch.empty(B, T, 2, H, **f32)
        dbi = [_grad_buf(cfg, b_ih_f), _grad_buf(cfg, b_ih_r)]
        dbh = [_grad_buf(cfg, b_hh_f), _grad_buf(cfg, b_hh_r)]
        # the recurrence also emits the bias gradients (column sums of dgi / dgh)
        bscr = torch.empty(2 * B * 6 * H, **f32)          # per-(clip, direction) bias-gradient records, summed in clip order
        # (records only: their sum -- sed_gru_bias_reduce -- feeds nothing but the optimizer and runs beside the chain, below)
        lib.call("sed_gru_bwd", dout.data_ptr(), out.data_ptr(), saved.data_ptr(), w_hh_f.data_ptr(), w_hh_r.data_ptr(),
                 dgi.data_ptr(), dgh.data_ptr(), hprev.data_ptr(), None, None, None, None, B, T, H, bscr.data_ptr(), st)

        bptr = (dbi[0].data_ptr(), dbi[1].data_ptr(), dbh[0].data_ptr(), dbh[1].data_ptr())

        def bias_sums(stream_ptr):
            lib.call("sed_gru_bias_reduce", bscr.data_ptr(), bptr[0], bptr[1], bptr[2], bptr[3], B, H, stream_ptr)
        if not DEFER_OFF_CHAIN:
            bias_sums(st)
        BT = B * T
        split = max(1, min(32, BT // 256))
        dwi = [_grad_buf(cfg, w_ih_f), _grad_buf(cfg, w_ih_r)]
        dwh = [_grad_buf(cfg, w_hh_f), _grad_buf(cfg, w_hh_r)]
        off = 3 * H * 4
        dx = None
        if ctx.needs_input_grad[0]:
            # the critical path first.  dx = [dgi_fwd | dgi_rev] . [W_ih_fwd ; W_ih_rev]: dgi rows already hold both directions
            # side by side, so this is ONE product over K = 6H whose B operand switches tensors at k = 3H (no atomics, no
            # zero fill)
            dx = torch.empty(B, T, I, **f32)
            kcat = "sed_gemm_kcat_bf16x3" if gemm_entry(cfg).endswith("bf16x3") else "sed_gemm_kcat"
            # one slice is (I / 64) x (B T / 128) workgroups walking 6 H / 32 dependent K tiles with the chip half empty: aim at ~ 700
            # workgroups (three resident per CU), slices of at least four tiles
            nsl = min(max(1, round(700.0 / (((I + 63) // 64) * ((BT + 127) // 128)))), max(1, (6 * H) // 128)) if GRU_DX_SPLITK else 1
            if kcat.endswith("bf16x3") and nsl > 1 and I % 4 == 0 and dx.data_ptr() % 16 == 0:
                scr = torch.empty(int(lib.value("sed_gemm_splitk_scratch_floats", BT, I, 6 * H, nsl)) // 2, **f32)
                lib.call("sed_gemm_kcat_splitk_bf16x3", dgi.data_ptr(), w_ih_f.data_ptr(), w_ih_r.data_ptr(), dx.data_ptr(), BT, I, 6 * H,
                         3 * H, 6 * H, I, I, nsl, scr.data_ptr(), st)
            else:
                lib.call(kcat, dgi.data_ptr(), w_ih_f.data_ptr(), w_ih_r.data_ptr(), dx.data_ptr(), BT, I, 6 * H, 3 * H, 6 * H, I, I, st)
        # dW_ih[d] = dgi[d]^T . x   and   dW_hh[d] = dgh[d]^T . hprev[d]   (K = B*T, split-K, both directions per launch).
        ws = st
        side = side_stream(x.device) if GRU_DW_SIDE else None
        # (addresses, not tensors, go into anything that outlives this call: see defer_off_chain)
        wptr = ([t.data_ptr() for t in dwi], [t.data_ptr() for t in dwh], [t.numel() for t in dwi], [t.numel() for t in dwh])

        def side_section(stream_ptr):
            if DEFER_OFF_CHAIN:
                bias_sums(stream_ptr)
            BiGRULayerFn._weight_grads(lib, cfg, dgi, dgh, hprev, x, wptr, B, T, I, H, split, stream_ptr, f32)
        off_chain_ok = _arena_views(cfg, *dwi, *dwh, *dbi, *dbh)
        if not off_chain_ok:
            side_section(ws)                    # some gradient buffer is not the arena's: written before this node returns
        elif (side is not None or (SIDE_ON_CPU and GRU_DW_SIDE)) and DEFER_OFF_CHAIN:
            # what the nodes before this one parked (the head's sums, the other layer's side section) goes out now that THIS layer's
            # recurrence and dX product -- the chain -- are enqueued; this layer's own side section waits for the next node's
            flush_deferred(side)
            defer_off_chain(x.device, side_section, (dgi, dgh, hprev, x, bscr))
        elif side is not None:
            side.wait_stream(torch.cuda.current_stream(x.device))          # after the recurrence and the dX product were enqueued
            for t in (dgi, dgh, hprev, x, bscr):
                t.record_stream(side)
            with torch.cuda.stream(side):
                side_section(side.cuda_stream)
        else:
            side_section(ws)
        d_w_ih, d_w_hh, d_b_ih, d_b_hh = dwi, dwh, dbi, dbh
        return (dx, d_w_ih[0], d_w_hh[0], d_b_ih[0], d_b_hh[0], d_w_ih[1], d_w_hh[1], d_b_ih[1], d_b_hh[1], None)

    @staticmethod
    def _weight_grads(lib, cfg, dgi, dgh, hprev, x, wptr, B, T, I, H, split, ws, f32):
        """wptr = ([&dW_ih fwd, rev], [&dW_hh fwd, rev], their element counts x 2): the outputs by address (see defer_off_chain)."""
        dwi, dwh, ni, nh = wptr
        BT, off = B * T, 3 * H * 4
        if (gemm_entry(cfg).endswith("bf16x3") and I % 4 == 0 and H % 4 == 0 and all(a % 16 == 0 for a in dwi + dwh)
                and not (cfg or {}).get("gru_dw_atomic", GRU_DW_ATOMIC)):
            # deterministic split-K: dense per-slice partials + a fixed-order sum (no zero fill, no fp32 atomics: 5 M atomics on
            # 98 K addresses were most of these launches)
            scr = torch.empty(int(lib.value("sed_gemm_splitk_scratch_floats", 3 * H, max(I, H), BT, split)), **f32)
            lib.call("sed_gemm_pair_splitk_bf16x3", dgi.data_ptr(), dgi.data_ptr() + off, x.data_ptr(), x.data_ptr(),
                     dwi[0], dwi[1], 3 * H, I, BT, 6 * H, I, I, 1, 0, split, scr.data_ptr(), ws)
            lib.call("sed_gemm_pair_splitk_bf16x3", dgh.data_ptr(), dgh.data_ptr() + off, hprev.data_ptr(), hprev.data_ptr() + H * 4,
                     dwh[0], dwh[1], 3 * H, H, BT, 6 * H, 2 * H, H, 1, 0, split, scr.data_ptr(), ws)
        else:
            lib.call("sed_zero_buffers", dwi[0], ni[0], dwi[1], ni[1],
                     dwh[0], nh[0], dwh[1], nh[1], ws)    # split-K GEMMs accumulate
            lib.call(gemm_entry(cfg), dgi.data_ptr(), dgi.data_ptr() + off, x.data_ptr(), x.data_ptr(), None, None,
                     dwi[0], dwi[1], 3 * H, I, BT, 6 * H, I, I, 1, 0, split, 0, ws)
            lib.call(gemm_entry(cfg), dgh.data_ptr(), dgh.data_ptr() + off, hprev.data_ptr(), hprev.data_ptr() + H * 4, None, None,
                     dwh[0], dwh[1], 3 * H, H, BT, 6 * H, 2 * H, H, 1, 0, split, 0, ws)


class EmbCatFn(torch.autograd.Function):
    """Embedding fusion of CRNN.py:283-296 (aggregation_type "pool1d"): cat_tf(dropout(cat(x, pool1d(emb)))).
    x (B,T,C), emb (B,E,Te) frozen features -> (B,T,C).  One fused pooling/concat/dropout kernel (K14) + the K7 GEMMs."""

    @staticmethod
    def forward(ctx, x, emb, w, b, cfg):
        lib = _lib.get()
        x, emb, w = x.contiguous(), emb.contiguous().float(), w.contiguous()
        _lib.check_tensor(x, "embcat input")
        _lib.check_tensor(emb, "embeddings")
        B, T, C = x.shape
        if emb.dim() != 3 or emb.shape[0] != B:
            raise ValueError("embeddings must be (batch, embedding_size, frames)")
        E, Te = emb.shape[1], emb.shape[2]
        if w.shape != (C, C + E):
            raise ValueError(f"cat_tf.weight is {tuple(w.shape)}, expected {(C, C + E)}")
        thr24, dscale = dropout_params(cfg.get("dropout_p", 0.0) if cfg.get("apply_dropout", False) else 0.0)
        seed = cfg.get("seed", 0)
        seed = seed if isinstance(seed, _graph.DynSeed) else int(seed)
        st = _lib.stream_ptr(x)
        f32 = dict(device=x.device, dtype=torch.float32)
        z = torch.empty(B, T, C + E, **f32)
        tmask, mode = cfg.get("tmask"), int(cfg.get("mode", 0))       # dropstep_recurrent bounds (B,4) int32; 1 = "interpolate"
        lib.call("sed_embcat_fwd", x.data_ptr(), emb.data_ptr(), z.data_ptr(), B, T, Te, C, E, int(seed), thr24, dscale,
                 _graph.seed_dev(seed), _p(tmask), mode, st)
        y = torch.empty(B, T, C, **f32)
        lib.call(gemm_entry(cfg, pair=False), z.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B * T, C, C + E, C + E, C + E, C,
                 0, 1, 1, 0, st)
        ctx.save_for_backward(z, w, b)
        ctx.tmask = tmask
        ctx.meta = (B, T, C, E, seed, thr24, dscale)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        z, w, b = ctx.saved_tensors
        B, T, C, E, seed, thr24, dscale = ctx.meta
        cfg = ctx.cfg
        dy = dy.contiguous()
        st = _lib.stream_ptr(dy)
        M, W = B * T, C + E
        entry = gemm_entry(cfg, pair=False)
        dx = None
        if ctx.needs_input_grad[0]:
            dzx = torch.empty(B, T, C, device=dy.device, dtype=torch.float32)
            lib.call(entry, dy.data_ptr(), w.data_ptr(), None, dzx.data_ptr(), M, C, C, C, W, C, 0, 0, 1, 0, st)   # dy . W[:, :C]
            dx = torch.empty_like(dzx)
            lib.call("sed_embcat_bwd", dzx.data_ptr(), dx.data_ptr(), M, C, E, int(seed), thr24, dscale, _graph.seed_dev(seed),
                     _p(ctx.tmask), T, st)
        dw, db = _grad_buf(cfg, w), _grad_buf(cfg, b)
        split = max(1, min(32, M // 256))
        if entry.endswith("bf16x3") and W % 4 == 0 and dw.data_ptr() % 16 == 0:
            # deterministic split-K (dense per-slice partials + a fixed-order sum): no zero fill, no float atomics
            scr = torch.empty(int(lib.value("sed_gemm_splitk_scratch_floats", C, W, M, split)), device=dy.device, dtype=torch.float32)
            lib.call("sed_gemm_splitk_bf16x3", dy.data_ptr(), z.data_ptr(), dw.data_ptr(), C, W, M, C, W, W, 1, 0, split, scr.data_ptr(), st)
        else:
            lib.call("sed_zero_buffers", dw.data_ptr(), dw.numel(), None, 0, None, 0, None, 0, st)             # split-K accumulates
            lib.call(entry, dy.data_ptr(), z.data_ptr(), None, dw.data_ptr(), C, W, M, C, W, W, 1, 0, split, 0, st)  # dy^T . z
        lib.call("sed_colsum", dy.data_ptr(), db.data_ptr(), None, C, M, C, C, st)
        return dx, None, dw, db, None


class DropStepFn(torch.autograd.Function):
    """`dropstep_recurrent` of a CRNN without embeddings (CRNN.py:296-301): dropout(time_mask(x)) on (B,T,C) in one kernel.
    bounds (B,2) int32 [t0, t1) or None."""

    @staticmethod
    def forward(ctx, x, bounds, cfg):
        x = x.contiguous()
        _lib.check_tensor(x, "dropstep input")
        B, T, C = x.shape
        thr24, dscale = dropout_params(cfg.get("dropout_p", 0.0) if cfg.get("apply_dropout", False) else 0.0)
        seed = cfg.get("seed", 0)
        seed = seed if isinstance(seed, _graph.DynSeed) else int(seed)
        y = torch.empty_like(x)
        _lib.get().call("sed_dropstep", x.data_ptr(), y.data_ptr(), _p(bounds), B, T, C, int(seed), thr24, dscale, _graph.seed_dev(seed),
                        _lib.stream_ptr(x))
        ctx.bounds = bounds
        ctx.meta = (B, T, C, seed, thr24, dscale)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, C, seed, thr24, dscale = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.get().call("sed_dropstep", dy.data_ptr(), dx.data_ptr(), _p(ctx.bounds), B, T, C, int(seed), thr24, dscale,
                        _graph.seed_dev(seed), _lib.stream_ptr(dy))
        return dx, None, None


class HeadFn(torch.autograd.Function):
    """Dropout + the two dense layers + class-softmax attention pooling (CRNN.py:152-178, :304).
    x (B,T,256) -> strong (B,T,NC) [caller exposes the (B,NC,T) view], weak (B,NC)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, cfg):
        lib = _lib.get()
        x = x.contiguous()
        _lib.check_tensor(x, "head input")
        B, T, D = x.shape
        NC = w1.shape[0]
        thr24, dscale = dropout_params(cfg.get("dropout_p", 0.0) if cfg.get("apply_dropout", False) else 0.0)
        seed = cfg.get("seed", 0)
        seed = seed if isinstance(seed, _graph.DynSeed) else int(seed)
        f32 = dict(device=x.device, dtype=torch.float32)
        strong = torch.empty(B, T, NC, **f32)
        psoft = torch.empty(B, T, NC, **f32)
        weak = torch.empty(B, NC, **f32)
        den = torch.empty(B, NC, **f32)
        w1, w2 = w1.contiguous(), w2.contiguous()
        cvalid, pad = cfg.get("classes_valid"), cfg.get("pad_mask")     # (B,NC) / (B,T) uint8 or None (CRNN.py:157-176)
        lib.call("sed_head_fwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), strong.data_ptr(),
                 psoft.data_ptr(), weak.data_ptr(), den.data_ptr(), B, T, D, NC, int(seed), thr24, dscale, _graph.seed_dev(seed),
                 _p(cvalid), _p(pad), _lib.stream_ptr(x))
        ctx.masks = (cvalid, pad)
        ctx.save_for_backward(x, w1, w2, strong, psoft, weak, den, b1, b2)
        ctx.meta = (B, T, D, NC, seed, thr24, dscale)
        ctx.cfg = cfg
        return strong, weak

    @staticmethod
    def backward(ctx, d_strong, d_weak):
        lib = _lib.get()
        probe("head_bwd")
        x, w1, w2, strong, psoft, weak, den, b1, b2 = ctx.saved_tensors
        B, T, D, NC, seed, thr24, dscale = ctx.meta
        cfg = ctx.cfg
        f32 = dict(device=x.device, dtype=torch.float32)
        d_strong = torch.zeros(B, T, NC, **f32) if d_strong is None else d_strong.contiguous()
        d_weak = torch.zeros(B, NC, **f32) if d_weak is None else d_weak.contiguous()
        dx = torch.empty_like(x)
        dw1, dw2 = _grad_buf(cfg, w1), _grad_buf(cfg, w2)
        db1, db2 = _grad_buf(cfg, b1), _grad_buf(cfg, b2)
        scratch = torch.empty(int(lib.value("sed_head_bwd_scratch_floats", B, T, D, NC)), device=x.device, dtype=torch.float32)
        # dx now; the weight / bias gradients (sums of the per-workgroup records) feed nothing but the optimizer: beside the chain
        lib.call("sed_head_bwd", x.data_ptr(), w1.data_ptr(), w2.data_ptr(), strong.data_ptr(), psoft.data_ptr(), weak.data_ptr(),
                 den.data_ptr(), d_strong.data_ptr(), d_weak.data_ptr(), dx.data_ptr(), None, None, None, None,
                 B, T, D, NC, int(seed), thr24, dscale, _graph.seed_dev(seed), _p(ctx.masks[0]),
                 _p(ctx.masks[1]), scratch.data_ptr(), _lib.stream_ptr(x))
        # (the parked launch holds ADDRESSES of the gradient buffers, not the tensors: a second reference would keep autograd's
        # AccumulateGrad from adopting them as .grad -- it would clone the not yet written buffers instead)
        outs = (dw1.data_ptr(), dw2.data_ptr(), db1.data_ptr(), db2.data_ptr())
        reduce = lambda stream_ptr: lib.call("sed_head_bwd_reduce", scratch.data_ptr(), outs[0], outs[1], outs[2], outs[3],    # noqa: E731
                                             B, T, D, NC, stream_ptr)
        if _arena_views(cfg, dw1, dw2, db1, db2):
            defer_off_chain(x.device, reduce, (scratch,))
        else:
            reduce(_lib.stream_ptr(x))
        return dx, dw1, db1, dw2, db2, None


class MeanTeacherLossFn(torch.autograd.Function):
    """Fused losses of SEDTask4.training_step (sed_trainer.py:309-342).
    Returns (scalars, total): scalars (8,) = [bce_strong, bce_weak, bce_strong_teacher, bce_weak_teacher, mse_strong, mse_weak,
    weight * (mse_strong + mse_weak), total] (not differentiable) and total = bce_strong + bce_weak + weight * (mse_strong +
    mse_weak) as a 0-d differentiable tensor -- all computed by the kernel (no scalar tensor arithmetic on the host side).
    selfsup_bce: slots 4 / 5 hold BCELoss(student, teacher) instead (`training.self_sup_loss: bce`).
    selfsup_from / valid: the 2024 multi-data-set step -- consistency terms over clips [selfsup_from, B), labels of classes
    outside a clip's data set (valid (B,NC) uint8 == 0) count as 0."""

    @staticmethod
    def forward(ctx, strong_s, weak_s, strong_t, weak_t, labels, labels_weak, n_strong, n_weak, weight, selfsup_bce=False,
                selfsup_from=0, valid=None):
        lib = _lib.get()
        strong_s, weak_s = strong_s.contiguous(), weak_s.contiguous()
        strong_t, weak_t = strong_t.contiguous(), weak_t.contiguous()
        labels, labels_weak = labels.contiguous().float(), labels_weak.contiguous().float()
        _lib.check_tensor(strong_s, "strong preds")
        B, T, NC = strong_s.shape
        f32 = dict(device=strong_s.device, dtype=torch.float32)
        buf = torch.empty(16, **f32)
        scalars, total = buf[:8], buf[8]            # two views of one buffer: the kernel writes the total into slots 7 and 8
        g_strong = torch.empty(B, T, NC, **f32)
        g_weak = torch.empty(B, NC, **f32)
        work = loss_work(strong_s.device, B)
        parked = AFTER_FORWARD is not None and (strong_s.is_cuda or SIDE_ON_CPU)
        # parked: only the gradient seeds and the per-clip records here -- the eight sums feed the log, not the backward pass, and go
        # out beside the chain (launcher.StepDriver.training_step_and_ema, next to the EMA)
        lib.call("sed_mt_loss_records" if parked else "sed_mt_loss", strong_s.data_ptr(), weak_s.data_ptr(), strong_t.data_ptr(),
                 weak_t.data_ptr(), labels.data_ptr(), labels_weak.data_ptr(), buf.data_ptr(), g_strong.data_ptr(), g_weak.data_ptr(),
                 B, T, NC, int(n_strong), int(n_weak), float(weight), getattr(weight, "dev", None), int(bool(selfsup_bce)),
                 int(selfsup_from), _p(valid), work.data_ptr(), _lib.stream_ptr(strong_s))
        if parked:
            AFTER_FORWARD.append((lambda stream_ptr: lib.call("sed_mt_loss_finish", work.data_ptr(), buf.data_ptr(), B, stream_ptr), (buf,)))
        ctx.save_for_backward(g_strong, g_weak)
        ctx.mark_non_differentiable(scalars)
        ctx.set_materialize_grads(False)
        return scalars, total

    @staticmethod
    def backward(ctx, _g_scalars, g_total):
        g_strong, g_weak = ctx.saved_tensors
        if g_total is None:
            return (None,) * 12
        if is_unit_grad(g_total):                   # loss.backward() through launcher.StepDriver: d(total) = 1, nothing to scale
            return g_strong, g_weak, None, None, None, None, None, None, None, None, None, None
        return g_strong * g_total, g_weak * g_total, None, None, None, None, None, None, None, None, None, None


_UNIT = {}
_LOSS_WORK = {}


def unit_grad(device):
    """A persistent 0-d tensor holding 1.0: `torch.autograd.backward(loss, unit_grad(dev))` instead of `loss.backward()` saves the
    ones_like fill and lets MeanTeacherLossFn.backward skip two elementwise multiplies by 1 (recognised by its storage)."""
    device = torch.device(device)
    key = (device.type, device.index)
    if key not in _UNIT:
        _UNIT[key] = torch.ones((), device=device, dtype=torch.float32)
    return _UNIT[key]


def is_unit_grad(g):
    u = _UNIT.get((g.device.type, g.device.index))
    return u is not None and g.data_ptr() == u.data_ptr() and g.dim() == 0


def loss_work(device, B):
    """Scratch of sed_mt_loss: per-clip partial sums + the ticket word (zeroed once: the kernel leaves the ticket at 0).  One buffer
    per (device, batch size, STREAM): launches on one stream are serialised, so everything that shares a buffer is ordered; two steps
    in flight on different streams (two tasks driven from two drivers) get their own tickets.  GraphedStepDriver creates the buffer
    before its capture begins, so it never lives in a graph's private pool."""
    device = torch.device(device)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device.type, device.index, int(B), stream)
    if key not in _LOSS_WORK:
        _LOSS_WORK[key] = torch.zeros(8 * B + 1, device=device, dtype=torch.float32)
    return _LOSS_WORK[key]


def reset_loss_work(device=None):
    """Zero the cached loss scratch (its ticket word in particular): a launch that faulted mid-way would otherwise leave a stale ticket
    behind for every later step.  Called when a step driver is constructed -- for ITS device, after synchronising it: the entries are
    keyed per stream, and zeroing another driver's scratch while its step is in flight would race with that step's ticket (ADVICE
    r04).  device None: every entry (tests)."""
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    for key, t in _LOSS_WORK.items():
        if device is None or (key[0], key[1]) == (torch.device(device).type, torch.device(device).index):
            t.zero_()
