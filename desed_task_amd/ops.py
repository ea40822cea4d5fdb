"""torch.autograd bindings of the CNN / GRU / head kernels (C-ABI in include/sed_hip.h).

PyTorch here is plumbing: it owns the HBM buffers, the stream and the autograd tape; every FLOP of the
path runs in libsed_hip.so.  Activations are channels-last (B, T, F, C) fp32.
"""
import torch

from . import _lib

BN_EPS = 1e-3        # desed_task/nnet/CNN.py:76
BN_MOMENTUM = 0.99


def _p(t):
    return None if t is None else t.data_ptr()


def dropout_params(p):
    """-> (thr24, dscale): keep element e iff (hash(e, seed) >> 8) >= thr24."""
    if p <= 0.0:
        return 0, 1.0
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return int(round(p * (1 << 24))), 1.0 / (1.0 - p)


def new_seed(generator=None):
    """A fresh 31-bit dropout seed from torch's CPU generator (host side, no device sync)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,), generator=generator).item())


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
        seed = int(cfg.get("seed", 0))
        bounds = cfg.get("bounds") if first else None
        st = _lib.stream_ptr(x)
        dev = x.device
        y = torch.empty(B, T, F, COUT, device=dev, dtype=torch.float32)
        nblk = lib.value("sed_conv_fwd_blocks", B, T, F, CIN)
        partial = torch.empty(nblk * 2 * COUT, device=dev, dtype=torch.float32) if training else None
        conv_w = conv_w.contiguous()
        if first:
            lib.call("sed_conv0_fwd", x.data_ptr(), conv_w.data_ptr(), _p(conv_b), _p(bounds), y.data_ptr(), _p(partial),
                     B, T, F, COUT, st)
        else:
            wf = torch.empty(9 * CIN * COUT, device=dev, dtype=torch.float32)
            lib.call("sed_conv_pack_weights", conv_w.data_ptr(), wf.data_ptr(), None, COUT, CIN, st)
            lib.call("sed_conv3x3", x.data_ptr(), wf.data_ptr(), _p(conv_b), y.data_ptr(), _p(partial), B, T, F, CIN, COUT, st)
        stats = torch.empty(4 * COUT, device=dev, dtype=torch.float32)
        lib.call("sed_bn_finalize", _p(partial), nblk, COUT, float(B * T * F), bn_w.data_ptr(), bn_b.data_ptr(),
                 running_mean.data_ptr(), running_var.data_ptr(), BN_MOMENTUM, BN_EPS, stats.data_ptr(), int(training),
                 int(training and cfg.get("update_running", True)), st)
        out = torch.empty(B, T // PT, F // PF, COUT, device=dev, dtype=torch.float32)
        glu_w = glu_w.contiguous()
        lib.call("sed_glu_fwd", y.data_ptr(), stats.data_ptr(), glu_w.data_ptr(), glu_b.data_ptr(), out.data_ptr(), B, T, F, COUT,
                 PT, PF, seed, thr24, dscale, st)
        ctx.save_for_backward(x, y, stats, conv_w, bn_w, bn_b, glu_w, glu_b)
        ctx.meta = (first, B, T, F, CIN, COUT, PT, PF, training, seed, thr24, dscale, bounds)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.get()
        x, y, stats, conv_w, bn_w, bn_b, glu_w, glu_b = ctx.saved_tensors
        first, B, T, F, CIN, COUT, PT, PF, training, seed, thr24, dscale, bounds = ctx.meta
        gout = gout.contiguous()
        dev = y.device
        st = _lib.stream_ptr(y)
        f32 = dict(device=dev, dtype=torch.float32)
        dz = torch.empty_like(y)
        d_glu_w = torch.empty(COUT, COUT, **f32)
        d_glu_b = torch.empty(COUT, **f32)
        d_gamma = torch.empty(COUT, **f32)
        d_beta = torch.empty(COUT, **f32)
        lib.call("sed_glu_bwd", y.data_ptr(), stats.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(), glu_w.data_ptr(), glu_b.data_ptr(),
                 gout.data_ptr(), dz.data_ptr(), d_glu_w.data_ptr(), d_glu_b.data_ptr(), d_gamma.data_ptr(), d_beta.data_ptr(),
                 B, T, F, COUT, PT, PF, seed, thr24, dscale, st)
        d_bias = torch.empty(COUT, **f32)
        lib.call("sed_bn_bwd_apply", y.data_ptr(), dz.data_ptr(), stats.data_ptr(), bn_w.data_ptr(), d_gamma.data_ptr(),
                 d_beta.data_ptr(), d_bias.data_ptr(), B * T * F, COUT, int(training), st)
        dy = dz
        d_w = torch.empty_like(conv_w)
        dx = None
        if first:
            lib.call("sed_conv0_wgrad", x.data_ptr(), _p(bounds), dy.data_ptr(), d_w.data_ptr(), B, T, F, COUT, st)
        else:
            scratch = torch.empty(9 * CIN * COUT, **f32)
            lib.call("sed_conv_wgrad", x.data_ptr(), dy.data_ptr(), scratch.data_ptr(), d_w.data_ptr(), B, T, F, CIN, COUT, st)
            if ctx.needs_input_grad[0]:
                wd = torch.empty(9 * CIN * COUT, **f32)
                lib.call("sed_conv_pack_weights", conv_w.data_ptr(), scratch.data_ptr(), wd.data_ptr(), COUT, CIN, st)
                dx = torch.empty_like(x)
                lib.call("sed_conv3x3", dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, B, T, F, COUT, CIN, st)
        return dx, d_w, d_bias, d_gamma, d_beta, d_glu_w, d_glu_b, None, None, None
