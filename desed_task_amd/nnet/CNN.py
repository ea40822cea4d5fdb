"""Parameter containers of the CNN encoder (state-dict layout of desed_task/nnet/CNN.py:5-114).

These modules only HOLD parameters/buffers under the reference's names (`cnn.conv{i}`, `cnn.batchnorm{i}`,
`cnn.glu{i}.linear`) and in the reference's construction order, so (a) published checkpoints load unchanged
and (b) a given torch seed yields the reference's initial weights.  They never run torch arithmetic: the
forward of every block is desed_task_amd.ops.ConvBlockFn (HIP kernels).
"""
import os

import torch
import torch.nn as nn

from .. import graph as _graph
from ..ops import ConvBlockFn, new_seed, pack_conv_weights


class GLU(nn.Module):
    """Holds Linear(C, C) of the gated unit `linear(x) * sigmoid(x)` (CNN.py:5-16)."""

    def __init__(self, input_num):
        super().__init__()
        self.sigmoid = nn.Sigmoid()
        self.linear = nn.Linear(input_num, input_num)

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("GLU is fused into the HIP CNN block; call CNN.forward")


class CNN(nn.Module):
    SUPPORTED = "activation='glu', normalization='batch', 3x3/stride 1/pad 1 convs, 1 input channel"

    def __init__(self, n_in_channel, activation="Relu", conv_dropout=0, kernel_size=[3, 3, 3], padding=[1, 1, 1],
                 stride=[1, 1, 1], nb_filters=[64, 64, 64], pooling=[(1, 4), (1, 4), (1, 4)], normalization="batch",
                 **transformer_kwargs):
        super().__init__()
        self.nb_filters = nb_filters
        # arithmetic of the 3x3 convolutions of blocks 1..: "f32" = exact-f32 MFMA, "bf16x3" = split-bf16 MFMA
        # (three bf16 MFMAs per product, fp32-level accuracy; see csrc/sed_conv_bf16.hip)
        self.conv_precision = os.environ.get("SED_CONV_PRECISION", transformer_kwargs.get("conv_precision", "bf16x3"))
        if self.conv_precision not in ("f32", "bf16x3"):
            raise ValueError("conv_precision must be 'f32' or 'bf16x3'")
        self.n_in_channel = n_in_channel
        self.conv_dropout = conv_dropout
        self.pooling = [tuple(p) for p in pooling]
        n_layers = len(nb_filters)
        ok = (str(activation).lower() == "glu" and normalization == "batch" and n_in_channel == 1
              and all(int(k) == 3 for k in kernel_size[:n_layers]) and all(int(p) == 1 for p in padding[:n_layers])
              and all(int(s) == 1 for s in stride[:n_layers]) and all(p in ((2, 2), (1, 2)) for p in self.pooling))
        if not ok:
            raise NotImplementedError("HIP CNN encoder supports only: " + self.SUPPORTED)
        layers = nn.Sequential()
        chans = [n_in_channel] + list(nb_filters)
        for i in range(n_layers):
            layers.add_module("conv%d" % i, nn.Conv2d(chans[i], chans[i + 1], 3, 1, 1))
            layers.add_module("batchnorm%d" % i, nn.BatchNorm2d(chans[i + 1], eps=0.001, momentum=0.99))
            layers.add_module("glu%d" % i, GLU(chans[i + 1]))
            if conv_dropout is not None:
                layers.add_module("dropout%d" % i, nn.Dropout(conv_dropout))
            layers.add_module("pooling%d" % i, nn.AvgPool2d(self.pooling[i]))
        self.cnn = layers
        # BatchNorm2d.num_batches_tracked is unused by the arithmetic (momentum is not None); it is kept exact for
        # checkpoints but bumped on the host and only written to the buffers when a state dict is requested
        # (7 tiny device launches per forward otherwise).
        self._pending_batches = [0] * n_layers
        self.last_bounds = None
        self.register_state_dict_pre_hook(CNN._flush_batch_counters)

    @staticmethod
    def _flush_batch_counters(module, prefix, keep_vars):
        for i, n in enumerate(module._pending_batches):
            if n:
                bn = module.cnn._modules["batchnorm%d" % i]
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += n
                module._pending_batches[i] = 0

    def _pack_weights(self, device, need_dgrad, prologue=None):
        """Repack the conv weights of blocks 1.. into the kernels' layouts, all layers in ONE launch -> {layer: (Wf, Wd)}."""
        mods = self.cnn._modules
        layers = list(range(1, len(self.nb_filters)))
        packs = pack_conv_weights([mods["conv%d" % i].weight for i in layers], need_dgrad, self.conv_precision, prologue)
        return dict(zip(layers, packs))

    FUSE_PROLOGUE = True                # bench.py --no-cnn-prologue (A/B): separate bounds / pack launches + x.clone()

    def can_fuse_prologue(self, x):
        """The one-launch prologue (ops.pack_conv_weights(prologue=...)) exists for the split-bf16 packs of a CNN with >= 2 blocks."""
        return (CNN.FUSE_PROLOGUE and self.conv_precision == "bf16x3" and len(self.nb_filters) > 1 and x.is_contiguous()
                and x.dtype == torch.float32 and x.data_ptr() % 16 == 0)       # (the copy moves 16-byte words)

    def forward(self, x, bounds=None, arena=None, specaug=None, private_input=False):
        """x: (B, T, F) scaled log-mel (channels-last with C = 1).  Returns (B, T', F', C_last) channels-last.
        bounds: optional (B,4) int32 SpecAugment bands fused into the first conv's load.
        specaug: instead of `bounds`, the parameters of a seeded draw (features.specaug_request) -- made by the weight-pack launch.
        private_input: x's storage is rewritten before this forward's backward pass runs (the pipelined step's hand-over buffer):
        work on a private copy -- made by the weight-pack launch as well when that exists, else x.clone()."""
        mods = self.cnn._modules
        p_drop = float(self.conv_dropout or 0.0)
        prologue = None
        if (specaug is not None or private_input) and self.can_fuse_prologue(x):
            prologue = {}
            if specaug is not None:
                bounds = torch.empty(x.shape[0], 4, dtype=torch.int32, device=x.device)
                prologue["bounds"] = dict(specaug, out=bounds)
            if private_input:
                mine = torch.empty_like(x)
                prologue["copy"] = (x, mine)
                x = mine
        else:
            if specaug is not None:
                from .. import features
                bounds = features.specaug_bounds_from_request(x.shape[0], specaug, x.device)
            if private_input:
                x = x.clone()
        packed = self._pack_weights(x.device, need_dgrad=torch.is_grad_enabled(), prologue=prologue)
        self.last_bounds = bounds           # diagnostics / test recorders: the SpecAugment bands of the latest forward (or None)
        for i in range(len(self.nb_filters)):
            conv, bn, glu = mods["conv%d" % i], mods["batchnorm%d" % i], mods["glu%d" % i]
            drop = mods.get("dropout%d" % i)
            apply_drop = drop is not None and drop.training and p_drop > 0
            cfg = dict(pool=self.pooling[i], bn_training=bn.training, dropout_p=p_drop, apply_dropout=apply_drop,
                       seed=new_seed() if apply_drop else 0, bounds=bounds if i == 0 else None, update_running=True,
                       arena=arena, packed=packed.get(i), conv_precision=self.conv_precision)
            x = ConvBlockFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, glu.linear.weight, glu.linear.bias,
                                  bn.running_mean, bn.running_var, cfg)
            if bn.training:
                dyn = _graph.active()
                if dyn is not None:                 # hipGraph step: host bookkeeping is re-run before every replay
                    dyn.host(lambda i=i: self._pending_batches.__setitem__(i, self._pending_batches[i] + 1))
                else:
                    self._pending_batches[i] += 1
        return x
