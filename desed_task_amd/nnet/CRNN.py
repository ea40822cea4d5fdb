"""Drop-in for desed_task.nnet.CRNN.CRNN (desed_task/nnet/CRNN.py:11-323) on the MI355X kernels.

Same constructor keywords, same `forward(x, pad_mask=None, embeddings=None, classes_mask=None) ->
(strong (B,nclass,T//4), weak (B,nclass))`, same state-dict keys (62 parameter tensors + 21 BN buffers), same
`train()` quirk (returns None).  Differences, all internal: activations are channels-last in HBM, the
parameters are views into one flat arena (arena.py), SpecAugment is folded into the first conv's load, and the
whole forward/backward is a chain of HIP kernels (ops.py).

Embedding fusion: `use_embeddings=True, aggregation_type="pool1d"` (the 2023 "pretrained" / BEATs
configuration, CRNN.py:143-144, :283-296) is built: `cat_tf = Linear(C + embedding_size, C)` after the heads in the state
dict, `forward(x, embeddings=emb (B, embedding_size, Te))`.

Also built: `aggregation_type="interpolate"` (nearest-exact, CRNN.py:271-279, the same fused kernel), `dropstep_recurrent`
(CRNN.py:288-301: per-clip time spans of the recurrent stage's input zeroed, independently for the CNN features and the
embeddings), `classes_mask` / `pad_mask` of the multi-data-set recipes (CRNN.py:157-176) inside the head kernels.

Not built (they raise NotImplementedError): aggregation_type "frame" (a 512-unit BiGRU encoder over the embedding frames, used by
no recipe configuration) and "global" (in the reference itself this branch ends in an undefined `reshape_emb`, CRNN.py:249-262 +
:295), cnn_integration, multi-head nclass lists.
"""
import copy

import torch
import torch.nn as nn

from .. import features
from .. import ops as _ops
from ..arena import ParamArena
from ..ops import DropStepFn, EmbCatFn, HeadFn, new_seed
from .CNN import CNN
from .RNN import BidirectionalGRU


class CRNN(nn.Module):
    def __init__(self, n_in_channel=1, nclass=10, attention=True, activation="glu", dropout=0.5, train_cnn=True,
                 rnn_type="BGRU", n_RNN_cell=128, n_layers_RNN=2, dropout_recurrent=0, cnn_integration=False,
                 freeze_bn=False, use_embeddings=False, embedding_size=527, embedding_type="global",
                 frame_emb_enc_dim=512, aggregation_type="global", specaugm_t_p=0.2, specaugm_t_l=5, specaugm_f_p=0.2,
                 specaugm_f_l=10, dropstep_recurrent=0.0, dropstep_recurrent_len=5, specaugm_iid_masks=True, **kwargs):
        super().__init__()
        if cnn_integration:
            raise NotImplementedError("cnn_integration is not built (SURVEY 8f)")
        if use_embeddings and aggregation_type not in ("pool1d", "interpolate"):
            raise NotImplementedError("use_embeddings is built for aggregation_type 'pool1d' and 'interpolate' ('frame' needs a "
                                      "512-unit BiGRU encoder no recipe uses; 'global' is broken in the reference itself)")
        if rnn_type != "BGRU":
            raise NotImplementedError("Only BGRU supported for CRNN for now")
        if isinstance(nclass, (tuple, list)):
            if len(nclass) > 1:
                raise NotImplementedError("multi-head nclass lists are not supported")
            nclass = nclass[0]
        if attention not in (True, "legacy"):
            raise NotImplementedError("the HIP head implements the attention-pooling variant (attention=True)")
        self.n_in_channel, self.attention, self.cnn_integration = n_in_channel, attention, cnn_integration
        self.freeze_bn, self.use_embeddings = freeze_bn, use_embeddings
        self.embedding_type, self.aggregation_type = embedding_type, aggregation_type
        self.nclass = nclass
        self.dropstep_recurrent, self.dropstep_recurrent_len = dropstep_recurrent, dropstep_recurrent_len
        self.specaugm_t_p, self.specaugm_t_l = specaugm_t_p, specaugm_t_l
        self.specaugm_f_p, self.specaugm_f_l = specaugm_f_p, specaugm_f_l
        # torchaudio >= 2.1 draws one SpecAugment mask per clip for this call; <= 2.0.x shared one mask per batch
        self.specaugm_iid_masks = specaugm_iid_masks
        self.dropout_p = float(dropout)

        # construction order == reference (same RNG consumption -> same default initialisation)
        self.cnn = CNN(n_in_channel=n_in_channel, activation=activation, conv_dropout=dropout, **kwargs)
        self.train_cnn = train_cnn
        if not train_cnn:
            for p in self.cnn.parameters():
                p.requires_grad = False
        self.rnn = BidirectionalGRU(n_in=self.cnn.nb_filters[-1], n_hidden=n_RNN_cell, dropout=dropout_recurrent,
                                    num_layers=n_layers_RNN)
        self.dropout = nn.Dropout(dropout)
        self.dense = nn.Linear(n_RNN_cell * 2, nclass)
        self.sigmoid = nn.Sigmoid()
        self.dense_softmax = nn.Linear(n_RNN_cell * 2, nclass)
        self.softmax = nn.Softmax(dim=-1)
        if use_embeddings:                                                # CRNN.py:143-144
            nb_in = self.cnn.nb_filters[-1]
            self.cat_tf = nn.Linear(nb_in + embedding_size, nb_in)
        self._arena = None
        self._build_arena()
        # data parallelism (launcher.StepDriver): with split_backward the autograd graph is cut at the CNN output, so that
        # loss.backward() stops after the recurrent stage -- the point where the all-reduce of the BiGRU / head gradients can start
        # -- and backward_cnn() runs the CNN's backward under it
        self.split_backward = False
        self._cnn_boundary = None

    # ---- flat parameter arena -------------------------------------------------------------------
    def _build_arena(self):
        old = self._arena
        self._arena = ParamArena(list(self.parameters()))
        if old is not None:
            old.successor = self._arena         # an optimizer built on the old arena follows the chain (arena.FusedAdam)

    @property
    def arena(self):
        if self._arena is None or not self._arena.is_intact():
            self._build_arena()
        return self._arena

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() go through here.  When nothing moved (Lightning calls model.to(device) at the start of fit
        # and test on a module that is already there) the parameters are still the arena's views: keep it, so that an
        # optimizer / captured graph holding its buffers stays valid.  Otherwise rebuild on the new storage.
        out = super()._apply(fn, *args, **kwargs)
        a = self._arena
        if a is None or not a.is_intact() or a.flat.device != a.params[0].device:
            self._build_arena()
        return out

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_arena", "_cnn_boundary"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._arena = None
        new._cnn_boundary = None
        new._build_arena()
        return new

    # ---- reference surface ----------------------------------------------------------------------
    def apply_specaugment(self, x):
        """Reference semantics on a (B, n_mels, T) tensor (CRNN.py:207-219); the forward() path fuses it instead."""
        if not self.training:
            return x
        b = self._specaug_bounds(x.shape[0], x.shape[1], x.shape[2], x.device)
        return features.specaug_apply(x, b) if b is not None else x

    def _specaug_bounds(self, B, n_freq, n_time, device):
        if min(self.specaugm_f_l, int(n_freq * self.specaugm_f_p)) < 1 and min(self.specaugm_t_l, int(n_time * self.specaugm_t_p)) < 1:
            return None
        # the uniforms come from the kernels' own counter-based generator, keyed by a host-drawn seed like the dropout masks (no
        # torch.rand launches in the step; `_ops.new_seed`, not the module-level name: test recorders count the DROPOUT sites)
        return features.specaug_bounds(B, n_freq, n_time, self.specaugm_f_l, self.specaugm_f_p, self.specaugm_t_l,
                                       self.specaugm_t_p, device, iid_masks=self.specaugm_iid_masks, seed=_ops.new_seed())

    def forward_cnn(self, x, private_input=False):
        """First half of forward(): SpecAugment + the 7 CNN blocks.  x (B, n_mels, T) -> (B, T', C) channels-last.
        private_input: the caller rewrites x's storage before the backward pass (see nnet/CNN.py: forward)."""
        if x.dim() != 3:
            raise ValueError("expected (batch, n_mels, frames)")
        xt = features.as_btf(x)                                           # (B, T, F), no copy for our own views
        request = None
        if self.training:
            # the seed is drawn here (the reference's order: apply_specaugment, then the CNN's dropouts); the draw itself happens in the
            # CNN's one-launch prologue next to the weight packs (`_ops.new_seed`, not the module-level name: test recorders count
            # the DROPOUT sites)
            request = features.specaug_request(xt.shape[0], xt.shape[2], xt.shape[1], self.specaugm_f_l, self.specaugm_f_p,
                                               self.specaugm_t_l, self.specaugm_t_p, self.specaugm_iid_masks, None)
            if request is not None:
                request["seed"] = _ops.new_seed()
        h = self.cnn(xt, arena=self.arena, specaug=request, private_input=private_input)      # (B, T', F', C)
        bs, frames, freq, chan = h.shape
        if freq != 1:
            raise NotImplementedError("CNN output keeps %d frequency bins; the recurrent stage expects 1" % freq)
        return h.view(bs, frames, chan)

    def forward_tail(self, h, embeddings=None, pad_mask=None, classes_mask=None):
        """Second half of forward(): [dropstep / embedding fusion +] BiGRU + dropout + attention head.
        (B, T', C) -> strong (B,nclass,T'), weak."""
        arena = self.arena
        if self.split_backward and h.requires_grad and torch.is_grad_enabled():
            cut = h.detach().requires_grad_(True)
            self._cnn_boundary = (h, cut)
            h = cut
            self.split_backward = False     # armed per step by launcher.StepDriver: a later plain loss.backward() is whole again
        B, Tp = h.shape[0], h.shape[1]
        dropstep = bool(self.dropstep_recurrent) and self.training                 # CRNN.py:288, :296
        if self.use_embeddings:
            if embeddings is None:
                raise ValueError("this CRNN was built with use_embeddings=True: forward() needs embeddings")
            if embeddings.requires_grad:
                raise NotImplementedError("embeddings are frozen features here (pretrained.e2e / unfrozen extractors are not built)")
            tmask = None
            if dropstep:        # two independent TimeMasking draws: the CNN features first, then the embeddings (:292-293)
                bx, be = self._dropstep_bounds(B, Tp, h.device), self._dropstep_bounds(B, Tp, h.device)
                tmask = torch.cat((bx, be), 1).contiguous() if bx is not None else None
            drop = self.dropout.training and self.dropout_p > 0
            cfg = dict(dropout_p=self.dropout_p, apply_dropout=drop, seed=new_seed() if drop else 0, arena=arena, tmask=tmask,
                       mode=1 if self.aggregation_type == "interpolate" else 0)
            h = EmbCatFn.apply(h, embeddings, self.cat_tf.weight, self.cat_tf.bias, cfg)
        elif embeddings is not None:
            raise ValueError("embeddings given to a CRNN built with use_embeddings=False")
        elif dropstep:          # x = dropout(dropstep(x)) -- this branch of the reference drops out the GRU input too (:296-301)
            drop = self.dropout.training and self.dropout_p > 0
            cfg = dict(dropout_p=self.dropout_p, apply_dropout=drop, seed=new_seed() if drop else 0)
            h = DropStepFn.apply(h, self._dropstep_bounds(B, Tp, h.device), cfg)
        h = self.rnn(h, arena=arena)                                      # (B, T', 256)
        drop = self.dropout.training and self.dropout_p > 0
        cfg = dict(dropout_p=self.dropout_p, apply_dropout=drop, seed=new_seed() if drop else 0, arena=arena,
                   classes_valid=self._byte_mask(classes_mask, (B, self.nclass), "classes_mask"),
                   pad_mask=self._byte_mask(pad_mask, (B, Tp), "pad_mask"))
        strong, weak = HeadFn.apply(h, self.dense.weight, self.dense.bias, self.dense_softmax.weight,
                                    self.dense_softmax.bias, cfg)
        return strong.transpose(1, 2), weak

    def _dropstep_bounds(self, B, n_time, device):
        """One torchaudio TimeMasking(dropstep_recurrent_len, iid_masks=True, p=dropstep_recurrent) draw over the frame axis
        (CRNN.py:289-291) -> (B,2) int32 [t0, t1), or None when the mask cannot be longer than 0."""
        if min(self.dropstep_recurrent_len, int(n_time * self.dropstep_recurrent)) < 1:
            return None
        b = features.specaug_bounds(B, 1, n_time, 0, 0.0, self.dropstep_recurrent_len, self.dropstep_recurrent, device,
                                    iid_masks=True, seed=_ops.new_seed())       # iid: hard-coded in the reference (CRNN.py:289-291, :297-299)
        return b[:, 2:4].contiguous()

    @staticmethod
    def _byte_mask(mask, shape, name):
        """bool / 0-1 mask of the reference call (`classes_mask` (B,nclass): True = class annotated in the clip's data set;
        `pad_mask` (B,1,T') or (B,T'): True = padded frame) -> contiguous uint8 tensor for the head kernels."""
        if mask is None:
            return None
        m = mask
        if name == "pad_mask" and m.dim() == 3:
            if m.shape[1] != 1:
                raise NotImplementedError("pad_mask must be (batch, 1, frames) or (batch, frames)")
            m = m[:, 0]
        if tuple(m.shape) != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (name, tuple(mask.shape), tuple(shape)))
        return (m != 0).to(torch.uint8).contiguous()

    def backward_cnn(self):
        """Second half of a split backward (split_backward = True): the CNN's backward from the gradient that loss.backward()
        left at the cut.  No-op when the last forward was not cut."""
        b, self._cnn_boundary = self._cnn_boundary, None
        if b is not None and b[1].grad is not None:
            b[0].backward(b[1].grad)

    def forward(self, x, pad_mask=None, embeddings=None, classes_mask=None):
        return self.forward_tail(self.forward_cnn(x), embeddings, pad_mask, classes_mask)

    def train(self, mode=True):
        """Mirrors CRNN.train (CRNN.py:308-323), including that it returns None."""
        super().train(mode)
        if self.freeze_bn:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
                    m.weight.requires_grad = False
                    m.bias.requires_grad = False
