"""Drop-in for the training step of the 2024 recipe's trainer, recipes/dcase2024_task4_baseline/local/sed_trainer_pretrained.py
(SURVEY 8f rank 3): the mean-teacher step over FIVE data sets per batch -- [MAESTRO | synthetic | strong real | weak |
unlabelled] -- with frozen embeddings, per-clip `valid_class_mask`, mixup inside each data set and consistency losses on
everything but MAESTRO (:318-430).

Same surface as the reference class for the step: constructor (with `pretrained_model`), `mel_spec`, `scaler`, `take_log`,
`detect(mel_feats, model, embeddings=None, **kwargs)`, `apply_mixup`, `training_step`, the EMA / scheduler hooks and the nine
logged keys.  Batches are `(audio, labels, padded_indxs, embeddings, valid_class_mask)` (:333-335).

Reference behaviours kept on purpose: the labels of a data-set group are mixed TWICE when embeddings are present -- once with
the features' (c, perm), once more with the embeddings' own draw (:283-301); the weak labels are derived after mixup (:354);
`train/student/tot_supervised` logs the strong consistency loss (:415); the consistency weight stops ramping at
`training.epoch_decay` (:393-396).

Not built from this file: validation / test of the 2024 recipe (MAESTRO segment metrics, class-wise median filters, mpAUC) and
`pretrained.e2e`.  The recipe's own `net:` section (n_RNN_cell 192, 27 classes, dropstep_recurrent) is what the parity fixtures use.
"""
import random

import numpy as np
import torch

from . import features
from . import graph as _graph
from .data_augm import MixupBatch, mixup_inplace_
from .ops import MeanTeacherLossFn
from .sed_trainer_pretrained import SEDTask4 as _SEDTask4


class SEDTask4(_SEDTask4):
    # `current_epoch` is LightningModule's read-only property under real Lightning; the stand-in base (sed_trainer._Base) carries
    # a plain attribute that a hand-written loop may set.  Nothing is defined here so that neither is shadowed.

    def detect(self, mel_feats, model, embeddings=None, **kwargs):
        x = self.scaled_logmel(mel_feats)
        if embeddings is None:
            return model(x, **kwargs)
        return model(x, embeddings=embeddings, **kwargs)

    def _unpack_batch(self, batch):
        if self.hparams["pretrained"]["e2e"]:
            raise NotImplementedError                       # as the reference (:306-316)
        if len(batch) != 5:
            raise ValueError("the 2024 step expects batches (audio, labels, padded_indxs, embeddings, valid_class_mask)")
        return batch

    def apply_mixup(self, features_, embeddings, labels, start_indx, stop_indx, dyn=None, gate=None, batches=(None, None)):
        """Mixup inside one data set, in place (:283-301): features + labels, then embeddings + labels again (second draw).
        batches: two MixupBatch objects -- the first-draw mixups of all data sets go out as one launch, the second-draw ones as
        another (the labels of a group are mixed by both draws, in this order)."""
        mixup_type = self.hparams["training"].get("mixup")
        sl = slice(start_indx, stop_indx)
        if stop_indx <= start_indx:
            return features_, embeddings, labels
        mixup_inplace_(features_[sl], labels[sl], mixup_label_type=mixup_type, dyn=dyn, gate=gate, batch=batches[0])
        if embeddings is not None:
            mixup_inplace_(embeddings[sl], labels[sl], mixup_label_type=mixup_type, dyn=dyn, gate=gate, batch=batches[1])
        return features_, embeddings, labels

    # ---- pipelined front half (SEDTask4.launch_prefetch, round 4) ---------------------------------------------------------------
    # The 2024 front half is mel -> mixup inside each data set (features + labels, then embeddings + labels again) -> weak labels
    # -> log / min-max.  Announced one step early it works on COPIES of the announced labels and embeddings (the hand-over buffers
    # `_pro["labels"]`, `_pro["embeddings"]`), so the caller's tensors are only read.
    prefetched_batch_fields = (1, 3)    # labels and embeddings come out of the hand-over buffers

    def next_batch_extras(self, next_batch):
        if len(next_batch) < 4 or next_batch[3] is None:
            raise ValueError("the 2024 step expects batches (audio, labels, padded_indxs, embeddings, valid_class_mask)")
        return {"embeddings": next_batch[3]}

    def _group_bounds(self):
        return tuple(int(v) for v in np.cumsum(self.hparams["training"]["batch_size"]))

    def _front_2024(self, audio, labels, embeddings, fresh=False, x_into_pro=False):
        """mel -> per-data-set mixup of (features, labels) and (embeddings, labels), in place on the tensors given -> weak labels
        (after mixup, :354) -> log / min-max (:318-356).  Host draws in the reference's order.  -> (x, labels_weak)"""
        features_ = self._features(audio, fresh)
        indx_maestro, indx_synth, indx_strong, indx_weak, indx_unlabelled = self._group_bounds()
        if indx_weak > features_.shape[0]:
            raise ValueError("batch smaller than the configured data-set sizes")
        mixup_type = self.hparams["training"].get("mixup")
        groups = ((indx_strong, indx_weak), (indx_maestro, indx_strong), (0, indx_maestro))          # :341-351, in this order
        dyn = _graph.active()
        if dyn is not None and mixup_type is not None:
            def flip():
                dyn.state["mixup"] = self.hparams["training"]["mixup_prob"] > random.random()
            dyn.host(flip)
            gate = lambda: dyn.state["mixup"]       # noqa: E731
            mbs = (MixupBatch(), MixupBatch())
            for a, b in groups:
                self.apply_mixup(features_, embeddings, labels, a, b, dyn=dyn, gate=gate, batches=mbs)
            mbs[0].launch()
            mbs[1].launch()
        elif mixup_type is not None and self.hparams["training"]["mixup_prob"] > random.random():
            mbs = (MixupBatch(), MixupBatch())
            for a, b in groups:
                self.apply_mixup(features_, embeddings, labels, a, b, batches=mbs)
            mbs[0].launch()
            mbs[1].launch()
        labels_weak = features.weak_labels(labels[indx_strong:indx_weak])       # after mixup (:354); class masking: loss kernel
        x_out = self._pro_buffer("x", features_) if x_into_pro else None
        return self.scaled_logmel(features_, out=x_out), labels_weak

    @staticmethod
    def _dense_embeddings(embeddings):
        embeddings = embeddings.float()
        return embeddings if embeddings.is_contiguous() else embeddings.contiguous()

    def _prefetch_front(self, audio, labels, extras):
        lab = self._pro_buffer("labels", labels)
        lab.copy_(labels)
        src = self._dense_embeddings(extras["embeddings"])
        emb = self._pro_buffer("embeddings", src)
        emb.copy_(src)
        return self._front_2024(audio, lab, emb, fresh=True, x_into_pro=True)

    def _training_step(self, batch, batch_indx):
        audio, labels, padded_indxs, embeddings, valid_class_mask = self._unpack_batch(batch)
        indx_maestro, indx_synth, indx_strong, indx_weak, indx_unlabelled = self._group_bounds()
        valid = (valid_class_mask != 0).to(torch.uint8).contiguous()
        dyn = _graph.active()
        pro = self._pro if (self._pro is not None and self._pro["ready"]) else None
        if pro is not None:
            # the previous step ran this step's front half and the teacher's CNN forward under its backward
            pro["ready"] = False
            if pro["labels"].shape != labels.shape or pro["embeddings"].shape != embeddings.shape:
                raise RuntimeError("the prefetched front half does not match this batch's shape")
            x, ht = pro["x"], pro["ht"]               # (x: volatile, the student's CNN copies it in its prologue launch)
            labels, labels_weak, embeddings = pro["labels"], pro["labels_weak"], pro["embeddings"]
        else:
            embeddings = self._dense_embeddings(embeddings)
            x, labels_weak = self._front_2024(audio, labels, embeddings)
            ht = None
        strong_s, weak_s, strong_t, weak_t = self._forward_pair(x, ht, embeddings, volatile_x=pro is not None, classes_mask=valid)

        sched = self.scheduler["scheduler"]
        const_max = self.hparams["training"]["const_max"]
        decay = self.hparams["training"].get("epoch_decay", float("inf"))

        def weight_now():
            return const_max * sched._get_scaling_factor() if self.current_epoch < decay else const_max
        weight = dyn.scalar(dyn.F_LOSS_W, weight_now) if dyn is not None else weight_now()
        scalars, tot_loss = MeanTeacherLossFn.apply(strong_s.transpose(1, 2), weak_s, strong_t.transpose(1, 2), weak_t, labels,
                                                    labels_weak, indx_strong, indx_weak - indx_strong, weight, self.selfsup_bce,
                                                    indx_maestro, valid)
        loss_strong, loss_weak, _, _, strong_self, weak_self, tot_self_loss, _ = scalars.unbind(0)
        lr = lambda: self.opt.param_groups[-1]["lr"] if self.opt is not None else 0.0      # noqa: E731
        self.log("train/student/loss_strong", loss_strong.detach())
        self.log("train/student/loss_weak", loss_weak.detach())
        if dyn is not None:
            dyn.host(lambda: (self.log("train/step", sched.step_num, prog_bar=True), self.log("train/lr", lr(), prog_bar=True)))
        self.log("train/step", sched.step_num, prog_bar=True)
        self.log("train/student/tot_self_loss", tot_self_loss, prog_bar=True)
        self.log("train/weight", weight.tensor if dyn is not None else weight)
        self.log("train/student/tot_supervised", strong_self.detach(), prog_bar=True)      # sic (reference :415)
        self.log("train/student/weak_self_sup_loss", weak_self.detach())
        self.log("train/student/strong_self_sup_loss", strong_self.detach())
        self.log("train/lr", lr(), prog_bar=True)
        self.last_outputs = (strong_s, weak_s, strong_t, weak_t)
        return tot_loss

    def validation_step(self, batch, batch_indx):
        # under pl.Trainer this fires at the sanity check already: run the recipe with limit_val_batches=0, num_sanity_val_steps=0
        #  -- this class replaces the TRAINING step only
        raise NotImplementedError("validation / test of the 2024 recipe (MAESTRO metrics, class-wise median filters) are not built: "
                                  "use limit_val_batches=0 and num_sanity_val_steps=0")

    test_step = validation_step
