"""SEDTask4 with the training-step surface of recipes/dcase2023_task4_baseline/local/sed_trainer.py:24-365,
running on the MI355X kernels.

Kept from the reference: constructor signature, `mel_spec`, `scaler`, `take_log`, `detect`, `training_step`
(batch tuple in, 0-d loss tensor with grad out, the same 11 logged keys, in-place mutation of features/labels
by mixup, python/numpy/torch-CPU RNG consumption order), `on_before_zero_grad` -> `update_ema`,
`lr_scheduler_step`, `configure_optimizers`, `on_save_checkpoint`, `train_dataloader`.
Validation / test forward + decoding (sed_trainer.py:367-487, 602-720) and the epoch-end metrics (:489-600, 722-975) are the
SURVEY 8f rank 1-2 rows: `validation_step`, `validation_epoch_end`, `test_step`, `on_test_epoch_end` below.

It subclasses pytorch_lightning.LightningModule when Lightning is importable, else a minimal stand-in with
`hparams`/`log` so the step can be driven by desed_task_amd.launcher (one process per GPU).
"""
import os
import random
from copy import deepcopy

import torch

from . import features
from . import graph as _graph
from . import ops as _ops
from .arena import FusedAdam, ema_update_
from .lookahead import LookaheadLoader
from .data_augm import MixupBatch, mixup_inplace_
from .ops import MeanTeacherLossFn
from .utils.scaler import TorchScaler

try:  # pragma: no cover - Lightning is not installed in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    from ._lightning_standin import LightningModule as _Base
from ._lightning_standin import move_to_device as _move_to_device


def _tensors(batch):
    if torch.is_tensor(batch):
        yield batch
    elif isinstance(batch, (list, tuple)):
        for b in batch:
            yield from _tensors(b)
    elif isinstance(batch, dict):
        for b in batch.values():
            yield from _tensors(b)


class SEDTask4(_Base):
    def __init__(self, hparams, encoder, sed_student, opt=None, train_data=None, valid_data=None, test_data=None,
                 train_sampler=None, scheduler=None, fast_dev_run=False, evaluation=False, sed_teacher=None):
        super().__init__()
        self.hparams.update(hparams)
        self.encoder = encoder
        self.sed_student = sed_student
        self.sed_teacher = deepcopy(sed_student) if sed_teacher is None else sed_teacher
        self.opt = opt
        self.train_data, self.valid_data, self.test_data = train_data, valid_data, test_data
        self.train_sampler = train_sampler
        self.scheduler = scheduler
        self.fast_dev_run = fast_dev_run
        self.evaluation = evaluation
        self.num_workers = 1 if fast_dev_run else self.hparams["training"]["num_workers"]

        feat = self.hparams["feats"]
        self.mel_spec = features.MelSpectrogram(
            sample_rate=feat["sample_rate"], n_fft=feat["n_window"], win_length=feat["n_window"],
            hop_length=feat["hop_length"], f_min=feat["f_min"], f_max=feat["f_max"], n_mels=feat["n_mels"],
            window_fn=torch.hamming_window, wkwargs={"periodic": False}, power=1)

        for p in self.sed_teacher.parameters():
            p.detach_()

        sup = self.hparams["training"]["self_sup_loss"]
        if sup not in ("mse", "bce"):                       # sed_trainer.py:97-103
            raise NotImplementedError
        self.selfsup_bce = sup == "bce"
        self.scaler = self._init_scaler()
        # train_sed.py:199-201 hands in torch.optim.Adam(sed_student.parameters(), lr, betas=(0.9, 0.999)): same object, one-launch step
        if opt is not None and os.environ.get("SED_ADOPT_ADAM", "1") != "0":
            FusedAdam.adopt(opt, self.sed_student)

    # ---- feature pipeline ------------------------------------------------------------------------
    def _init_scaler(self):
        sc = self.hparams["scaler"]
        if sc["statistic"] == "instance":
            return TorchScaler("instance", sc["normtype"], sc["dims"])
        if sc["statistic"] != "dataset":
            raise NotImplementedError
        # sed_trainer.py:218-250: data-set statistics are loaded from `savepath` when it exists, else fitted on the log-mels of
        # one pass over the training loader (and saved).  Knowing deviation: the reference forgets to return the freshly fitted
        # scaler when `savepath` is None (falls off the end -> None); it is returned here.
        import os
        scaler = TorchScaler("dataset", sc["normtype"], sc["dims"])
        path = sc.get("savepath")
        if path is not None and os.path.exists(path):
            print("Loaded Scaler from previous checkpoint from {}".format(path))
            return torch.load(path, weights_only=False)
        self.train_loader = self.train_dataloader()
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        with torch.no_grad():               # the mel kernel has no host path: the clips go to the GPU for the pass
            scaler.fit(self.train_loader, transform_func=lambda x: self.take_log(self.mel_spec(x[0].to(dev))))
        if path is not None:
            torch.save(scaler, path)
            print("Saving Scaler from previous checkpoint at {}".format(path))
        return scaler

    def take_log(self, mels):
        return features.take_log(mels)

    def _fused_scaler(self):
        s = self.scaler
        return isinstance(s, TorchScaler) and s.statistic == "instance" and s.normtype == "minmax" and tuple(s.dims) == (1, 2)

    def scaled_logmel(self, mel_feats, out=None):
        """scaler(take_log(mels)) -- one fused log + per-clip min/max + affine pass when the scaler is instance/minmax.
        out: write into this tensor (mel_feats' shape and strides) when the fused pass runs; ignored otherwise."""
        if self._fused_scaler():
            return features.minmax_scale(mel_feats, eps=self.scaler.eps, apply_log=True, out=out)
        return self.scaler(self.take_log(mel_feats))

    def detect(self, mel_feats, model, embeddings=None):
        if embeddings is None:
            return model(self.scaled_logmel(mel_feats))
        return model(self.scaled_logmel(mel_feats), embeddings=embeddings)      # sed_trainer_pretrained.py:276-280

    # ---- optimisation hooks -----------------------------------------------------------------------
    def lr_scheduler_step(self, scheduler, optimizer_idx=None, metric=None):
        if self._take("scheduler"):
            return
        dyn = _graph.active()
        if dyn is not None:
            dyn.host(scheduler.step)            # pure host arithmetic: re-run before every graph replay
        else:
            scheduler.step()

    def update_ema(self, alpha, global_step, model, ema_model):
        alpha = min(1 - 1 / (global_step + 1), alpha)
        ema_update_(list(ema_model.parameters()), list(model.parameters()), alpha,
                    getattr(ema_model, "arena", None), getattr(model, "arena", None))

    def on_before_zero_grad(self, *args, **kwargs):
        if self._take("ema"):
            return
        factor = self.hparams["training"]["ema_factor"]
        sched = self.scheduler["scheduler"]
        dyn = _graph.active()
        if dyn is not None:
            alpha = dyn.scalar(dyn.F_EMA_ALPHA, lambda: min(1 - 1 / (sched.step_num + 1), factor), complement=True)
            ema_update_(list(self.sed_teacher.parameters()), list(self.sed_student.parameters()), alpha,
                        getattr(self.sed_teacher, "arena", None), getattr(self.sed_student, "arena", None))
            return
        self.update_ema(factor, sched.step_num, self.sed_student, self.sed_teacher)

    def configure_optimizers(self):
        return [self.opt], [self.scheduler]

    def on_save_checkpoint(self, checkpoint):
        checkpoint["sed_student"] = self.sed_student.state_dict()
        checkpoint["sed_teacher"] = self.sed_teacher.state_dict()
        return checkpoint

    def train_dataloader(self):
        # sed_trainer.py:913-920's DataLoader, as the subclass that stays one batch ahead (lookahead.py): the whole-step mode below
        # learns the NEXT batch from it.  A data set of ready batches (lookahead.BatchList) is iterated as it is.
        if self.train_sampler is None and getattr(self.train_data, "yields_batches", False):
            self.train_loader = LookaheadLoader(self.train_data, batch_size=None, num_workers=0)
        else:
            self.train_loader = LookaheadLoader(self.train_data, batch_sampler=self.train_sampler, num_workers=self.num_workers)
        return self.train_loader

    def val_dataloader(self):               # sed_trainer.py:922-930
        self.val_loader = torch.utils.data.DataLoader(self.valid_data, batch_size=self.hparams["training"]["batch_size_val"],
                                                      num_workers=self.num_workers, shuffle=False, drop_last=False)
        return self.val_loader

    def test_dataloader(self):              # sed_trainer.py:932-940
        self.test_loader = torch.utils.data.DataLoader(self.test_data, batch_size=self.hparams["training"]["batch_size_val"],
                                                       num_workers=self.num_workers, shuffle=False, drop_last=False)
        return self.test_loader

    # ---- whole-step mode: the benchmarked launch path behind Lightning's own loop ---------------------------------------------------
    # `pl.Trainer.fit` (train_sed.py:278-299) drives one batch through   training_step -> on_before_zero_grad -> optimizer_zero_grad ->
    # backward -> optimizer.step -> lr_scheduler_step   (Lightning 1.9's closure).  Run hook by hook that is ~330 launches
    # from one Python thread plus a 62-tensor Adam: host-bound.  In whole-step mode `training_step` hands the batch to the step
    # driver -- graph.GraphedStepDriver on the GPU: the captured hipGraph of the WHOLE optimisation step (both forwards, the losses,
    # the EMA, backward, the one-launch Adam, the scheduler's host arithmetic), with the next batch's front half pipelined under this
    # batch's backward; launcher.StepDriver on a CPU device (the test emulator) -- and returns the (detached) loss; the hooks
    # Lightning calls afterwards for the same batch find their work done (`_take`) and return.  The optimizer Lightning steps is the
    # adopted FusedAdam: its step() runs the closure -- which is where training_step is called -- and skips the update once (`served`).
    # The next batch comes from `train_dataloader()`'s look-ahead loader; epoch ends, checkpoint loads and batches the loader does not
    # know fall back to eager launches / an inline front half inside the driver (GraphedStepDriver._run_step).  Same kernels, same
    # host draws in the same order: bit-identical to driving GraphedStepDriver by hand (tests/lightning_order.py).
    #   SED_WHOLE_STEP = 0 | 1 | auto (default: on when the student is on a GPU and nothing below blocks it);  `whole_step` overrides.
    whole_step = None
    whole_step_prefetch = None              # None: SED_PREFETCH or "teacher"; "off" | "tails" | "backward" | "teacher"
    whole_step_warmup = 3                   # eager steps before the capture (GraphedStepDriver)
    _driver = None
    _in_driver = False                      # launcher.StepDriver sets it around ITS call of training_step
    _served = None
    _static_logs = None
    _cur_batch = None                       # (loader key, device batch) noted by transfer_batch_to_device
    _uploaded = None                        # (loader key, device batch) of the batch announced as NEXT (uploaded once, used twice)
    _whole_ok = None

    def _take(self, what):
        """Has the whole-step driver already done `what` for the current batch?  (One-shot: a second call does the work.)"""
        s = self._served
        if s and what in s:
            s.discard(what)
            return True
        return False

    def _whole_step_blockers(self):
        from .nnet.CRNN import CRNN
        tr = self.hparams["training"]
        if not (isinstance(self.sed_student, CRNN) and isinstance(self.sed_teacher, CRNN)):
            return "student / teacher are not desed_task_amd.nnet.CRNN"
        if not isinstance(self.opt, FusedAdam) or self.scheduler is None:
            return "the optimizer is not a plain Adam over sed_student.parameters() (arena.FusedAdam.adopt), or there is no scheduler"
        if tr.get("accumulate_batches", 1) != 1:
            return "accumulate_batches != 1 (one optimizer step per training_step is what a whole step is)"
        if (tr.get("gradient_clip") or 0) > 0:
            return "gradient_clip > 0 (clipping sits between backward and the optimizer step)"
        if self._ddp_world() > 1 and os.environ.get("SED_DDP_GRAPH_EXCHANGE") != "1":
            # (with the exchange captured -- SED_DDP_GRAPH_EXCHANGE=1 -- the whole data-parallel step, all-reduce(s) and Adam included,
            #  is the one graph behind training_step and the hooks that follow find their work done, exactly as at world size 1.  Without
            #  it the exchange is host-driven between the replay and Adam, which belongs to desed_task_amd.launcher's own loop.)
            return ("a process group with more than one rank is up: set SED_DDP_GRAPH_EXCHANGE=1 (the gradient exchange becomes part of "
                    "the captured step) or run data-parallel through desed_task_amd.launcher")
        return None

    @staticmethod
    def _ddp_world():
        import torch.distributed as dist
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _whole_step_on(self):
        if self._whole_ok is None:
            mode = self.whole_step
            if mode is None:
                mode = {"0": False, "1": True}.get(os.environ.get("SED_WHOLE_STEP", "auto"))
            if mode is False:
                self._whole_ok = False
            else:
                why = self._whole_step_blockers()
                if why and mode is True:
                    raise RuntimeError("whole-step mode was asked for but cannot run: " + why)
                on_gpu = next(self.sed_student.parameters()).device.type == "cuda"
                self._whole_ok = (not why) and (mode is True or on_gpu)
        return self._whole_ok and self.training

    def _step_driver(self, pipelined):
        if self._driver is None:
            pf = self.whole_step_prefetch or os.environ.get("SED_PREFETCH", "teacher")
            if not pipelined:
                pf = "off"              # nobody can announce a next batch (no look-ahead loader): the unpipelined step
            elif pf == "teacher" and not self.prefetch_teacher_ok:
                pf = "backward"
            dev = next(self.sed_student.parameters()).device
            world = self._ddp_world()       # > 1: every rank's step exchanges its gradients (launcher.StepDriver: start-up broadcast,
            if dev.type == "cuda":          # mean over the flat arena, 1 / world folded into Adam)
                self._driver = _graph.GraphedStepDriver(self, world_size=world, warmup=self.whole_step_warmup, prefetch=pf)
                self._driver.eager.check_announced = False
            else:
                from .launcher import StepDriver
                self._driver = StepDriver(self, world_size=world, prefetch=pf)
                self._driver.check_announced = False    # this class tracks the identity of the batches itself (loader keys)
        return self._driver

    def _epoch_limit(self):
        """Batches the trainer runs per epoch when it is fewer than the loader holds (`limit_train_batches`, train_sed.py:256)."""
        try:
            n = getattr(self.trainer, "num_training_batches", None)
        except Exception:  # noqa: BLE001 -- real Lightning raises when no trainer is attached
            n = None
        return n if isinstance(n, int) else None

    # Host batches (what a DataLoader hands out): batch k + 1 must be on the device before step k's replay starts (its side branch reads
    # it), so uploading it at step k would put 30.7 MB of PCIe traffic in front of every replay.  Batch k + 2 is therefore uploaded
    # DURING step k, on an upload stream of its own (pinned memory: asynchronous; pageable: the host thread blocks, the GPU does not),
    # and step k + 1 merely waits for that upload's event.  Device-resident batches pass through untouched.
    _staged = None                          # {loader key: (device batch, upload-done event or None)}
    _up_stream = None

    def _stage(self, key, host_batch, device):
        st = self._staged
        if st is None:
            st = self._staged = {}
        for k in [k for k in st if k[0] != key[0] or k[1] < key[1] - 2]:
            del st[k]                       # (another epoch's, or batches nobody asked for)
        if key in st:
            return
        on_host = [t for t in _tensors(host_batch) if t.device != device]
        if not on_host or device.type != "cuda":
            st[key] = (_move_to_device(host_batch, device), None)
            return
        if self._up_stream is None:
            self._up_stream = torch.cuda.Stream(device=device)
        with torch.cuda.stream(self._up_stream):
            dev_batch = _move_to_device(host_batch, device)
            ev = torch.cuda.Event()
            ev.record()
        st[key] = (dev_batch, ev)

    def _take_staged(self, key, device):
        dev_batch, ev = self._staged.pop(key)
        if ev is not None:
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            for t in _tensors(dev_batch):
                if t.is_cuda:
                    t.record_stream(cur)    # allocated on the upload stream, read on this one
        return dev_batch

    def transfer_batch_to_device(self, batch, device, dataloader_idx=0):
        """Lightning's hook, called once per batch right before training_step.  A training batch of the look-ahead loader that was
        already uploaded -- as the announced successor of the previous batch, or one step further ahead -- is not uploaded again."""
        loader = getattr(self, "train_loader", None)
        if self.training and isinstance(loader, LookaheadLoader):
            key = loader.find(batch)
            if key is not None:
                up = self._uploaded
                if up is not None and up[0] == key:
                    dev_batch = up[1]
                elif self._staged and key in self._staged:
                    dev_batch = self._take_staged(key, torch.device(device))
                else:
                    dev_batch = _move_to_device(batch, device)
                self._cur_batch = (key, dev_batch)
                return dev_batch
        return super().transfer_batch_to_device(batch, device, dataloader_idx)

    def _next_from_loader(self, batch):
        """The batch that follows `batch` in the look-ahead loader's epoch, on the device -- or None (no such loader, a batch it does not
        know, the end of the epoch).  Also starts the upload of the batch after that."""
        loader = getattr(self, "train_loader", None)
        if not isinstance(loader, LookaheadLoader):
            return None
        cur = self._cur_batch
        key = cur[0] if (cur is not None and batch[0] is cur[1][0]) else loader.find(batch)
        self._cur_batch = None
        announced = self._uploaded[0] if self._uploaded is not None else None
        if announced is not None and key != announced:
            # The previous step announced -- and prefetched the front half of -- a batch that is not the one we were given: the epoch
            # was abandoned mid-way (a break / max_steps, then a new iter(loader)) or the caller skipped a batch.  What sits in the
            # hand-over buffers (features, mixed labels, the teacher's CNN output) belongs to a batch nobody trains on now: forget it,
            # this step runs its own front half inline (it used to be consumed silently -- the step trained on the old batch).
            self.reset_pipeline()
            self._uploaded = None
            self._staged = None
        if key is None:
            return None
        loader.release(key)
        limit = self._epoch_limit()
        device = batch[0].device
        k1, k2 = (key[0], key[1] + 1), (key[0], key[1] + 2)
        nxt = loader.batch_after(key) if (limit is None or k1[1] < limit) else None
        if nxt is None:
            self._uploaded = None
            return None
        self._stage(k1, nxt, device)
        dev_next = self._take_staged(k1, device)
        self._uploaded = (k1, dev_next)
        nxt2 = loader.batch_after(k1) if (limit is None or k2[1] < limit) else None
        if nxt2 is not None:
            self._stage(k2, nxt2, device)
        return dev_next

    def training_step(self, batch, batch_indx):
        if self._in_driver or not self._whole_step_on():
            return self._training_step(batch, batch_indx)
        nxt = self._next_from_loader(batch)
        drv = self._step_driver(pipelined=nxt is not None)
        self._served = None
        # (a caller that runs training_step twice without an optimizer.step() in between -- a hand-written loop, a bench -- must not
        #  leave the previous call's "update already applied" mark for the driver's OWN nested optimizer step to find: ADVICE r05)
        self.opt.served = False
        graph_before = getattr(drv, "graph", None)
        rec = []
        object.__setattr__(self, "log", lambda name, value, **kw: rec.append((name, value, kw)))    # (no self.log inside a capture)
        try:
            loss = drv.run_step(batch, batch_indx, next_batch=nxt)
        finally:
            object.__delattr__(self, "log")
        if any(name == "train/student/loss_strong" for name, _, _ in rec):
            # an eager or the capture step: every key was logged; after a capture the tensors are the graph's static outputs
            if graph_before is None and getattr(drv, "graph", None) is not None:
                self._static_logs = list(rec)
        elif self._static_logs is not None:
            # a replay ran no Python but the host half (train/step, train/lr): the rest are the static tensors it just rewrote
            fresh = {name: (value, kw) for name, value, kw in rec}
            rec = [(name,) + fresh.get(name, (value, kw)) for name, value, kw in self._static_logs]
        for name, value, kw in rec:
            self.log(name, value, **kw)
        self._served = {"ema", "zero_grad", "backward", "scheduler"}
        self.opt.served = True
        # "0-d loss tensor with grad" (SURVEY 8b): a leaf, so that Lightning's `loss / accumulate_grad_batches` and a hand-written
        # `loss.backward()` both work -- there is nothing left to differentiate
        return loss.detach().requires_grad_(True)

    def optimizer_zero_grad(self, epoch, batch_idx, optimizer, *args, **kwargs):
        if not self._take("zero_grad"):
            super().optimizer_zero_grad(epoch, batch_idx, optimizer, *args, **kwargs)

    def backward(self, loss, *args, **kwargs):
        if not self._take("backward"):
            super().backward(loss, *args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self.reset_pipeline()           # a front half prefetched with the previous weights must not be consumed (launcher.load_checkpoint)
        return out

    # ---- the hot path -----------------------------------------------------------------------------
    overlap_tails = True            # student / teacher GRU+head tails on two HIP streams (GPU only)
    _tstream = None

    def _tail_stream(self, device):
        from .nnet.CRNN import CRNN
        if (device.type != "cuda" or not self.overlap_tails or not isinstance(self.sed_student, CRNN)
                or not isinstance(self.sed_teacher, CRNN)):
            return None
        if self._tstream is None:
            self._tstream = torch.cuda.Stream(device=device)
        return self._tstream

    def _batch_embeddings(self, batch):
        """Frozen embeddings of the batch, or None: the plain recipe has none (sed_trainer.py:280 ignores batch[3])."""
        return None

    def _eval_embeddings(self, batch):
        """Same for validation / test batches (audio, labels, padded_indxs, filenames[, embeddings])."""
        return None

    # ---- software-pipelined front half (launcher.StepDriver(prefetch=...)) -------------------------------------------------
    # Nothing in the FRONT HALF of step k + 1 -- the mel kernel, mixup, the log / min-max pass -- depends on step k, and the
    # teacher's CNN forward of step k + 1 depends only on the teacher weights after step k's EMA, which runs before step k's
    # backward.  Meanwhile the back half of a step leaves most of the chip idle for long stretches: the BiGRU recurrences run 96
    # workgroups on 256 CUs.  When the driver announces the next batch (`set_next_batch`), `launch_prefetch()` -- called at the
    # fork point -- enqueues, on a side stream,
    #   level "features": the mel kernel of batch k + 1 into the persistent feature buffer `_feat_buf`;
    #   level "teacher" : the whole front half of step k + 1 -- mel, weak labels, mixup (features and labels, host draws in the
    #                     reference's order), log / min-max -- and the teacher's CNN forward, kept in `_pro`;
    # and the next training_step starts from there.  Same kernels, same inputs, same draws (the teacher's CNN draws its dropout /
    # SpecAugment seeds from its own private stream, ops.seed_stream("teacher_cnn"), so running it early changes no mask): results
    # are bit-identical to the unpipelined order.  Buffers: the feature buffer's only readers (mixup, log / min-max) precede the
    # fork in stream order; the step's own labels were consumed by the loss kernel before the fork; the scaled features x are
    # read by the student's backward AFTER the fork, so a prefetched x is cloned at the head of the step that consumes it.
    prefetch_point = None               # None (off) | "tails" (fork before the BiGRU + head tails) | "backward" (before backward)
    prefetch_level = "features"         # "features" | "teacher" (needs prefetch_point == "backward": after the EMA)
    _feat_buf = None
    _feat_ready = False
    _next_audio = None
    _next_labels = None
    _next_extras = None                 # other tensors of the announced batch the front half needs (2024: the embeddings)
    _pro = None
    _pf_stream = None

    def set_next_audio(self, audio):
        """Announce the waveforms of the NEXT batch (None: there is none); consumed by launch_prefetch() in this step."""
        self._next_audio = audio

    def set_next_batch(self, audio, labels=None, extras=None):
        """Announce the NEXT batch: waveforms and -- for prefetch_level "teacher" -- its labels (copied into the hand-over buffer and
        mixed THERE one step early; the caller's tensor is only read) and whatever else of it the front half needs (`extras`, see
        next_batch_extras)."""
        self._next_audio, self._next_labels, self._next_extras = audio, labels, (extras or None)

    prefetch_teacher_ok = True          # this class's training_step consumes a "teacher"-level prefetch (subclasses that override
                                        # training_step without that support must set it to False)

    prefetched_batch_fields = (1,)      # batch-tuple positions (besides the waveforms) a primed "teacher"-level step does not read

    def next_batch_extras(self, next_batch):
        """{name: tensor} of the announced batch tuple that the pipelined front half reads besides waveforms and labels.  The plain
        and the 2023 `pretrained` steps need nothing (their embeddings enter behind the CNN and are not mixed)."""
        return {}

    def _feature_buffer(self, audio):
        T = 1 + audio.shape[1] // self.mel_spec.hop_length
        shape = (audio.shape[0], T, self.mel_spec.n_mels)
        if self._feat_buf is None or tuple(self._feat_buf.shape) != shape or self._feat_buf.device != audio.device:
            if self._feat_ready:
                raise RuntimeError("the batch shape changed between a prefetch and the step that consumes it")
            self._feat_buf = torch.empty(shape, device=audio.device, dtype=torch.float32)
        return self._feat_buf

    def launch_prefetch(self, point, after=(), fork_event=None):
        """Fork point `point` of the step: if it is the configured one and a next batch was announced, enqueue its front half on
        the prefetch stream (ordered after everything the current stream -- and the streams in `after`: the EMA's -- has enqueued
        so far)."""
        audio, labels, extras = self._next_audio, self._next_labels, self._next_extras
        if point != self.prefetch_point or audio is None:
            return
        self._next_audio = self._next_labels = self._next_extras = None
        teacher = self.prefetch_level == "teacher" and labels is not None
        if teacher and not self.prefetch_teacher_ok:
            raise NotImplementedError('prefetch_level "teacher" is not built for %s.training_step' % type(self).__name__)
        if teacher and point != "backward":
            raise RuntimeError('prefetch_level "teacher" needs the fork point "backward" (the teacher weights after this step\'s EMA)')

        def body():
            _ops.probe("prefetch_start")
            if not teacher:
                self.mel_spec.frames_major(audio, out=self._feature_buffer(audio))
                self._feat_ready = True
                return
            x, lab_w = self._prefetch_front(audio, labels, extras or {})
            with torch.no_grad(), _ops.seed_stream("teacher_cnn"):
                ht = self.sed_teacher.forward_cnn(x)
            # into PERSISTENT buffers: a captured step reads fixed addresses (x was written there directly when the scaler is fused)
            self._pro_buffer("labels_weak", lab_w).copy_(lab_w)
            px = self._pro_buffer("x", x)
            if px.data_ptr() != x.data_ptr():
                px.copy_(x)
            self._pro_buffer("ht", ht).copy_(ht)
            self._pro["ready"] = True
            _ops.probe("prefetch_end")

        if audio.device.type != "cuda":
            body()
            return
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=audio.device)
        main = torch.cuda.current_stream(audio.device)
        if fork_event is not None:
            self._pf_stream.wait_event(fork_event)      # fork at an EARLIER point of the current stream than where we are now
        else:
            self._pf_stream.wait_stream(main)
        for s in after:
            if s is not None:
                self._pf_stream.wait_stream(s)
        # the announced tensors were allocated on the caller's stream and may be released by the caller as soon as run_step returns:
        # tell the allocator that this stream still reads (and mixes) them
        for t in (audio, labels) + tuple((extras or {}).values()):
            if t is not None:
                t.record_stream(self._pf_stream)
        with torch.cuda.stream(self._pf_stream):
            body()

    def _prefetch_front(self, audio, labels, extras):
        """Front half of the ANNOUNCED batch for the pipelined step -> (x, labels_weak); the mixed labels are left in the hand-over
        buffer `_pro["labels"]`.  They are mixed THERE, not in the caller's tensor: an inline front half that has to re-run on the
        same announced batch (after reset_pipeline(): weights loaded in between) then starts from unmixed labels."""
        lab = self._pro_buffer("labels", labels)
        lab.copy_(labels)
        x, _, lab_w = self._front(audio, lab, fresh=True, x_into_pro=True)
        return x, lab_w

    def _pro_buffer(self, key, like):
        """Persistent hand-over buffer `key` of the pipelined front half, shaped like `like` (created on first use)."""
        p = self._pro
        if p is None:
            p = self._pro = {"ready": False}
        t = p.get(key)
        if t is None or t.shape != like.shape or t.device != like.device or t.dtype != like.dtype or (
                t.stride() != like.stride() and t.stride() != torch.empty_like(like).stride()):
            # (the second stride test: a non-dense `like` -- sliced labels -- gets a dense buffer from empty_like; comparing it with
            #  like's own strides would reallocate on every call)
            if p["ready"]:
                raise RuntimeError("the batch shape changed between a prefetch and the step that consumes it")
            if p.get("frozen") and t is not None:
                raise RuntimeError("hand-over buffer %r changed shape / layout after the step was captured into a hipGraph: the replays "
                                   "read the old address (build a new GraphedStepDriver for another batch shape)" % key)
            t = p[key] = torch.empty_like(like)      # (preserve_format: x is a (B, n_mels, T) VIEW of frame-major storage and must stay one)
        return t

    def reset_pipeline(self):
        """Forget a front half that was prefetched for the next step (after weights were loaded in between: the teacher's CNN output
        in the hand-over buffers belongs to the old weights).  The next step then runs its front half inline."""
        self._feat_ready = False
        self._next_audio = self._next_labels = self._next_extras = None
        if self._pro is not None:
            self._pro["ready"] = False

    def join_prefetch(self):
        """The current stream waits for the prefetch stream (end of the step: the next step reads what it produced, and a
        capture must not end with a forked stream still open)."""
        if self._pf_stream is not None:
            torch.cuda.current_stream(self._pf_stream.device).wait_stream(self._pf_stream)

    def _features(self, audio, fresh=False):
        """Linear mels (B, n_mels, T) of this batch: the prefetched buffer when the previous step computed them, else the kernel
        now (into the feature buffer when the pipelined front-end is on, so that a captured graph always reads one address)."""
        if self._feat_ready and not fresh:
            self._feat_ready = False
            buf = self._feat_buf
            if buf.shape[0] != audio.shape[0] or buf.shape[1] != 1 + audio.shape[1] // self.mel_spec.hop_length:
                raise RuntimeError("prefetched features do not match this batch's shape")
            return buf.transpose(1, 2)
        if self.prefetch_point is not None:
            return self.mel_spec.frames_major(audio, out=self._feature_buffer(audio)).transpose(1, 2)
        return self.mel_spec(audio)

    def _front(self, audio, labels, fresh=False, x_into_pro=False):
        """Front half of the step (sed_trainer.py:280-301 + the log / scaler part of detect): mel -> weak labels -> mixup of the
        weak and the strong group (features and labels in place; coin flip, c, permutations drawn on the host in the reference's
        order) -> log + per-clip min-max.  -> (x, labels, labels_weak)"""
        indx_synth, indx_weak, indx_unlabelled = self.hparams["training"]["batch_size"]
        features_ = self._features(audio, fresh)                          # (B, n_mels, T) view of frame-major HBM
        batch_num = features_.shape[0]
        if indx_synth + indx_weak > batch_num:
            raise ValueError("batch smaller than the configured strong+weak sizes")
        weak_sl = slice(indx_synth, indx_synth + indx_weak)
        strong_sl = slice(0, indx_synth)
        labels_weak = features.weak_labels(labels[weak_sl])             # (sum over frames > 0).float(), one launch

        mixup_type = self.hparams["training"].get("mixup")
        dyn = _graph.active()
        if dyn is not None and mixup_type is not None:
            # hipGraph step: the mixup launches are always part of the graph; the coin flip, c and the permutations are
            # host draws (same order as below) re-run before every replay and read by the kernels from device memory
            def flip():
                dyn.state["mixup"] = 0.5 > random.random()
            dyn.host(flip)
            gate = lambda: dyn.state["mixup"]       # noqa: E731
            mb = MixupBatch()                       # weak features + weak labels + strong features + strong labels: one launch
            mixup_inplace_(features_[weak_sl], labels_weak, mixup_label_type=mixup_type, dyn=dyn, gate=gate, batch=mb)
            mixup_inplace_(features_[strong_sl], labels[strong_sl], mixup_label_type=mixup_type, dyn=dyn, gate=gate, batch=mb)
            mb.launch()
        elif mixup_type is not None and 0.5 > random.random():
            mb = MixupBatch()
            mixup_inplace_(features_[weak_sl], labels_weak, mixup_label_type=mixup_type, batch=mb)
            mixup_inplace_(features_[strong_sl], labels[strong_sl], mixup_label_type=mixup_type, batch=mb)
            mb.launch()
        # x is shared by student and teacher; the pipelined front half writes it straight into its hand-over buffer
        x_out = self._pro_buffer("x", features_) if x_into_pro else None
        return self.scaled_logmel(features_, out=x_out), labels, labels_weak

    def _forward_pair(self, x, ht=None, embeddings=None, volatile_x=False, **tail_kw):
        """Student (with grad) and teacher (no grad) forward on the same scaled features x -> (strong_s, weak_s, strong_t, weak_t).
        ht: the teacher's CNN output when the previous step already computed it (pipelined front half), else None.
        tail_kw: forward_tail keywords of the multi-data-set recipes (classes_mask, pad_mask).
        volatile_x: x is the pipeline's hand-over buffer -- this step's backward still reads the features while the next prefetch
        rewrites it, so the student works on a private copy (made by its CNN's one-launch prologue; it was a clone() launch).
        Both CNN encoders first (they fill the GPU), then the two latency-bound tails -- BiGRU recurrence (96 workgroups each) + head --
        side by side on two HIP streams: they are independent and together still leave CUs idle.  (Running the WHOLE teacher forward
        concurrently was measured to be a net loss.)  The teacher's CNN draws its dropout / SpecAugment seeds from its own private
        stream, so that when it runs -- here, or one step ahead -- changes no mask."""
        from .nnet.CRNN import CRNN
        split = isinstance(self.sed_student, CRNN) and isinstance(self.sed_teacher, CRNN)
        tstream = self._tail_stream(x.device)
        if volatile_x and not split:
            x, volatile_x = x.clone(), False
        if tstream is None:
            self.launch_prefetch("tails")       # (single-stream / CPU path: the position of the fork is immaterial)
            if split:
                strong_s, weak_s = self.sed_student.forward_tail(self.sed_student.forward_cnn(x, private_input=volatile_x), embeddings, **tail_kw)
            else:
                strong_s, weak_s = self.sed_student(x, embeddings=embeddings, **tail_kw)
            with torch.no_grad():
                if split:
                    if ht is None:
                        with _ops.seed_stream("teacher_cnn"):
                            ht = self.sed_teacher.forward_cnn(x)
                    strong_t, weak_t = self.sed_teacher.forward_tail(ht, embeddings, **tail_kw)
                else:
                    strong_t, weak_t = self.sed_teacher(x, embeddings=embeddings, **tail_kw)
            return strong_s, weak_s, strong_t, weak_t
        hs = self.sed_student.forward_cnn(x, private_input=volatile_x)
        if ht is None:
            with torch.no_grad(), _ops.seed_stream("teacher_cnn"):
                ht = self.sed_teacher.forward_cnn(x)
        main = torch.cuda.current_stream(x.device)
        self.launch_prefetch("tails")
        tstream.wait_stream(main)
        with torch.cuda.stream(tstream), torch.no_grad():
            strong_t, weak_t = self.sed_teacher.forward_tail(ht, embeddings, **tail_kw)
        strong_s, weak_s = self.sed_student.forward_tail(hs, embeddings, **tail_kw)
        main.wait_stream(tstream)
        ht.record_stream(tstream)
        strong_t.record_stream(main)
        weak_t.record_stream(main)
        return strong_s, weak_s, strong_t, weak_t

    def _training_step(self, batch, batch_indx):
        """sed_trainer.py:269-356: the step body (what `training_step` is when the hooks run one by one)."""
        audio, labels = batch[0], batch[1]
        embeddings = self._batch_embeddings(batch)        # NOT mixed up with the features (sed_trainer_pretrained.py:320-330)
        indx_synth, indx_weak, indx_unlabelled = self.hparams["training"]["batch_size"]
        dyn = _graph.active()
        pro = self._pro if (self._pro is not None and self._pro["ready"]) else None
        if pro is not None:
            # the previous step ran this step's front half and the teacher's CNN forward under its backward
            pro["ready"] = False
            if pro["labels"].shape != labels.shape:
                raise RuntimeError("the prefetched front half does not match this batch's shape")
            x, ht = pro["x"], pro["ht"]
            labels, labels_weak = pro["labels"], pro["labels_weak"]
        else:
            x, labels, labels_weak = self._front(audio, labels)
            ht = None
        strong_s, weak_s, strong_t, weak_t = self._forward_pair(x, ht, embeddings, volatile_x=pro is not None)
        sched = self.scheduler["scheduler"]
        const_max = self.hparams["training"]["const_max"]
        if dyn is not None:
            weight = dyn.scalar(dyn.F_LOSS_W, lambda: const_max * sched._get_scaling_factor())
        else:
            weight = const_max * sched._get_scaling_factor()
        scalars, tot_loss = MeanTeacherLossFn.apply(strong_s.transpose(1, 2), weak_s, strong_t.transpose(1, 2), weak_t, labels,
                                                    labels_weak, indx_synth, indx_weak, weight, self.selfsup_bce)
        loss_strong, loss_weak, loss_strong_t, loss_weak_t, strong_self, weak_self, tot_self_loss, _ = scalars.unbind(0)

        self.log("train/student/loss_strong", loss_strong.detach())
        self.log("train/student/loss_weak", loss_weak.detach())
        self.log("train/teacher/loss_strong", loss_strong_t.detach())
        self.log("train/teacher/loss_weak", loss_weak_t.detach())
        if dyn is not None:     # host-side scalars: refreshed before every replay (the device-side ones are static tensors)
            dyn.host(lambda: (self.log("train/step", sched.step_num, prog_bar=True),
                              self.log("train/lr", self.opt.param_groups[-1]["lr"] if self.opt is not None else 0.0,
                                       prog_bar=True)))
        self.log("train/step", sched.step_num, prog_bar=True)
        self.log("train/student/tot_self_loss", tot_self_loss, prog_bar=True)
        self.log("train/weight", weight.tensor if dyn is not None else weight)
        self.log("train/student/tot_supervised", strong_self.detach(), prog_bar=True)      # sic (reference :351)
        self.log("train/student/weak_self_sup_loss", weak_self.detach())
        self.log("train/student/strong_self_sup_loss", strong_self.detach())
        self.log("train/lr", self.opt.param_groups[-1]["lr"] if self.opt is not None else 0.0, prog_bar=True)
        self.last_outputs = (strong_s, weak_s, strong_t, weak_t)
        return tot_loss

    # ---- validation forward + decoding ---------------------------------------------
    class _MacroF1:
        """Stand-in for torchmetrics MultilabelF1Score(num_labels, average="macro") at threshold 0.5 (sed_trainer.py:106-118):
        per-class tp / fp / fn counters kept on the device."""

        def __init__(self):
            self.tp = self.fp = self.fn = None

        def __call__(self, preds, target):
            p, t = preds > 0.5, target > 0
            tp, fp, fn = (p & t).sum(0), (p & ~t).sum(0), (~p & t).sum(0)
            if self.tp is None:
                self.tp, self.fp, self.fn = tp, fp, fn
            else:
                self.tp, self.fp, self.fn = self.tp + tp, self.fp + fp, self.fn + fn

        def compute(self):
            if self.tp is None:             # no weak clip seen (e.g. limit_val_batches cut them off): torchmetrics answers 0 as well
                return torch.zeros(())
            den = 2 * self.tp + self.fp + self.fn
            return torch.where(den > 0, 2.0 * self.tp / den.clamp(min=1), torch.zeros_like(den, dtype=torch.float32)).mean()

        def reset(self):
            self.tp = self.fp = self.fn = None

    def _val_state(self):
        """Buffers of sed_trainer.py:106-135, created on first use (the training-only configurations carry no val keys)."""
        if not hasattr(self, "val_buffer_student_synth"):
            import pandas as pd
            ths = self.hparams["training"].get("val_thresholds", [0.5])
            self.val_buffer_student_synth = {k: pd.DataFrame() for k in ths}
            self.val_buffer_teacher_synth = {k: pd.DataFrame() for k in ths}
            self.val_scores_postprocessed_buffer_student_synth = {}
            self.val_scores_postprocessed_buffer_teacher_synth = {}
            self.get_weak_student_f1_seg_macro = SEDTask4._MacroF1()
            self.get_weak_teacher_f1_seg_macro = SEDTask4._MacroF1()

    def validation_step(self, batch, batch_indx):
        """sed_trainer.py:367-487 with the per-clip host loop of `batched_decode_preds` replaced by the batched device
        post-processing (desed_task_amd/postprocess.py).  Same logged keys, same buffers.  The epoch-end metrics
        (PSDS / intersection / event F1, :489-600) are the next row and are not computed here."""
        from pathlib import Path
        import pandas as pd
        from .postprocess import batched_decode_preds
        self._val_state()
        audio, labels, padded_indxs, filenames = batch[0], batch[1], batch[2], batch[3]
        bce = torch.nn.functional.binary_cross_entropy
        emb = self._eval_embeddings(batch)
        with torch.no_grad():
            mels = self.mel_spec(audio)
            x = self.scaled_logmel(mels)                     # student and teacher see the same features (detect() twice)
            strong_s, weak_s = self.sed_student(x, embeddings=emb)
            strong_t, weak_t = self.sed_teacher(x, embeddings=emb)
        data = self.hparams.get("data", {})
        weak_dir, synth_dir = data.get("weak_folder"), data.get("synth_val_folder")
        is_weak = [weak_dir is not None and str(Path(f).parent) == str(Path(weak_dir)) for f in filenames]
        is_synth = [synth_dir is not None and str(Path(f).parent) == str(Path(synth_dir)) for f in filenames]
        mask_weak = torch.tensor(is_weak, device=audio.device)
        mask_synth = torch.tensor(is_synth, device=audio.device)
        if any(is_weak):
            labels_weak = (torch.sum(labels[mask_weak], -1) >= 1).float()
            self.log("val/weak/student/loss_weak", bce(weak_s[mask_weak], labels_weak))
            self.log("val/weak/teacher/loss_weak", bce(weak_t[mask_weak], labels_weak))
            self.get_weak_student_f1_seg_macro(weak_s[mask_weak], labels_weak.long())
            self.get_weak_teacher_f1_seg_macro(weak_t[mask_weak], labels_weak.long())
        if any(is_synth):
            self.log("val/synth/student/loss_strong", bce(strong_s[mask_synth], labels[mask_synth]))
            self.log("val/synth/teacher/loss_strong", bce(strong_t[mask_synth], labels[mask_synth]))
            filenames_synth = [f for f, s in zip(filenames, is_synth) if s]
            win = self.hparams["training"].get("median_window", 7)
            for preds, buf, post in ((strong_s, self.val_buffer_student_synth, self.val_scores_postprocessed_buffer_student_synth),
                                     (strong_t, self.val_buffer_teacher_synth, self.val_scores_postprocessed_buffer_teacher_synth)):
                _, scores_post, decoded = batched_decode_preds(preds[mask_synth], filenames_synth, self.encoder,
                                                               median_filter=win, thresholds=list(buf.keys()))
                post.update(scores_post)
                for th in buf.keys():
                    buf[th] = pd.concat([buf[th], decoded[th]], ignore_index=True)
        return

    def _scored_ground_truth(self, tsv, dur, score_buffer):
        """Ground-truth / duration dicts for the threshold-free PSDS as the reference prepares them (sed_trainer.py:503-525,
        :736-758): fast_dev_run keeps the scored clips, otherwise clips without events are dropped."""
        from .evaluation.psds_scores import read_audio_durations, read_ground_truth_events
        ground_truth, audio_durations = read_ground_truth_events(tsv), read_audio_durations(dur)
        if self.fast_dev_run:
            ground_truth = {a: ground_truth[a] for a in score_buffer}
        else:
            ground_truth = {a: gt for a, gt in ground_truth.items() if len(gt) > 0}
        return ground_truth, {a: audio_durations[a] for a in ground_truth}

    def validation_epoch_end(self, outputs=None):
        """sed_trainer.py:489-600 on this package's evaluators (desed_task_amd/evaluation): same objective
        selection (`training.obj_metric_synth_type`: None / "psds" -> threshold-free PSDS scenario 1, "event", "intersection"),
        same logged keys, same buffer resets."""
        import pandas as pd
        from .evaluation.evaluation_measures import (compute_per_intersection_macro_f1, compute_psds_from_scores,
                                                     log_sedeval_metrics)
        self._val_state()
        obj_type = self.hparams["training"].get("obj_metric_synth_type")
        if obj_type not in (None, "psds", "event", "intersection"):
            raise NotImplementedError(f"obj_metric_synth_type: {obj_type} not implemented.")
        data = self.hparams["data"]
        weak_student_f1_macro = self.get_weak_student_f1_seg_macro.compute()
        weak_teacher_f1_macro = self.get_weak_teacher_f1_seg_macro.compute()
        ground_truth, audio_durations = self._scored_ground_truth(data["synth_val_tsv"], data["synth_val_dur"],
                                                                  self.val_scores_postprocessed_buffer_student_synth)
        psds1_student_sed_scores_eval = compute_psds_from_scores(
            self.val_scores_postprocessed_buffer_student_synth, ground_truth, audio_durations, dtc_threshold=0.7,
            gtc_threshold=0.7, cttc_threshold=None, alpha_ct=0, alpha_st=1)
        intersection_f1_macro_student = compute_per_intersection_macro_f1(self.val_buffer_student_synth, data["synth_val_tsv"],
                                                                          data["synth_val_dur"])
        synth_student_event_macro = log_sedeval_metrics(self.val_buffer_student_synth[0.5], data["synth_val_tsv"])[0]
        intersection_f1_macro_teacher = compute_per_intersection_macro_f1(self.val_buffer_teacher_synth, data["synth_val_tsv"],
                                                                          data["synth_val_dur"])
        synth_teacher_event_macro = log_sedeval_metrics(self.val_buffer_teacher_synth[0.5], data["synth_val_tsv"])[0]
        if obj_type in (None, "psds"):
            synth_metric = psds1_student_sed_scores_eval
        elif obj_type == "event":
            synth_metric = synth_student_event_macro
        else:
            synth_metric = intersection_f1_macro_student
        obj_metric = torch.tensor(float(weak_student_f1_macro) + float(synth_metric))
        self.log("val/obj_metric", obj_metric, prog_bar=True)
        self.log("val/weak/student/macro_F1", weak_student_f1_macro)
        self.log("val/weak/teacher/macro_F1", weak_teacher_f1_macro)
        self.log("val/synth/student/psds1_sed_scores_eval", psds1_student_sed_scores_eval)
        self.log("val/synth/student/intersection_f1_macro", intersection_f1_macro_student)
        self.log("val/synth/teacher/intersection_f1_macro", intersection_f1_macro_teacher)
        self.log("val/synth/student/event_f1_macro", synth_student_event_macro)
        self.log("val/synth/teacher/event_f1_macro", synth_teacher_event_macro)
        # free the buffers
        ths = self.hparams["training"].get("val_thresholds", [0.5])
        self.val_buffer_student_synth = {k: pd.DataFrame() for k in ths}
        self.val_buffer_teacher_synth = {k: pd.DataFrame() for k in ths}
        self.val_scores_postprocessed_buffer_student_synth = {}
        self.val_scores_postprocessed_buffer_teacher_synth = {}
        self.get_weak_student_f1_seg_macro.reset()
        self.get_weak_teacher_f1_seg_macro.reset()
        return obj_metric

    # ---- test scoring (sed_trainer.py:608-911) ----------------------------------------------------------
    _exp_dir = None

    @property
    def exp_dir(self):
        if self._exp_dir is None:
            self._exp_dir = getattr(getattr(self, "logger", None), "log_dir", None) or self.hparams["log_dir"]
        return self._exp_dir

    def _test_state(self):
        """Buffers of sed_trainer.py:139-150, created on first use."""
        if not hasattr(self, "test_psds_buffer_student"):
            import numpy as np
            import pandas as pd
            n = self.hparams["training"]["n_test_thresholds"]
            test_thresholds = np.arange(1 / (n * 2), 1, 1 / n)
            self.test_psds_buffer_student = {k: pd.DataFrame() for k in test_thresholds}
            self.test_psds_buffer_teacher = {k: pd.DataFrame() for k in test_thresholds}
            self.decoded_student_05_buffer = pd.DataFrame()
            self.decoded_teacher_05_buffer = pd.DataFrame()
            self.test_scores_raw_buffer_student = {}
            self.test_scores_raw_buffer_teacher = {}
            self.test_scores_postprocessed_buffer_student = {}
            self.test_scores_postprocessed_buffer_teacher = {}

    def test_step(self, batch, batch_indx):
        """sed_trainer.py:608-683: student and teacher posteriors of one batch, median-filtered and decoded at the
        n_test_thresholds PSDS thresholds + 0.5 by the device post-processing (two launches and one copy per model instead
        of the reference's per-clip, per-threshold host loop)."""
        import pandas as pd
        from .postprocess import batched_decode_preds
        self._test_state()
        audio, labels, padded_indxs, filenames = batch[0], batch[1], batch[2], batch[3]
        emb = self._eval_embeddings(batch)
        with torch.no_grad():
            mels = self.mel_spec(audio)
            x = self.scaled_logmel(mels)
            strong_s, weak_s = self.sed_student(x, embeddings=emb)
            strong_t, weak_t = self.sed_teacher(x, embeddings=emb)
        if not self.evaluation:
            bce = torch.nn.functional.binary_cross_entropy
            self.log("test/student/loss_strong", bce(strong_s, labels))
            self.log("test/teacher/loss_strong", bce(strong_t, labels))
        win = self.hparams["training"].get("median_window", 7)
        for preds, psds_buf, raw, post, who in (
                (strong_s, self.test_psds_buffer_student, self.test_scores_raw_buffer_student,
                 self.test_scores_postprocessed_buffer_student, "student"),
                (strong_t, self.test_psds_buffer_teacher, self.test_scores_raw_buffer_teacher,
                 self.test_scores_postprocessed_buffer_teacher, "teacher")):
            scores_raw, scores_post, decoded = batched_decode_preds(preds, filenames, self.encoder, median_filter=win,
                                                                    thresholds=list(psds_buf.keys()) + [0.5])
            raw.update(scores_raw)
            post.update(scores_post)
            for th in psds_buf.keys():
                psds_buf[th] = pd.concat([psds_buf[th], decoded[th]], ignore_index=True)
            if who == "student":
                self.decoded_student_05_buffer = pd.concat([self.decoded_student_05_buffer, decoded[0.5]])
            else:
                self.decoded_teacher_05_buffer = pd.concat([self.decoded_teacher_05_buffer, decoded[0.5]])

    def on_test_epoch_end(self):
        """sed_trainer.py:685-911.  `evaluation=True`: only the raw / post-processed score tables are written (one TSV per
        clip, the sed_scores_eval.io.write_sed_scores layout).  Otherwise PSDS scenario 1 / 2 from the operating points
        (psds_eval role) and from the score tables (sed_scores_eval role), event-based and intersection-based macro F1, for
        both models -- the reference's `test/...` keys.  Not reproduced: the codecarbon energy keys (trackers are outside the
        hot path)."""
        import os
        from .evaluation.evaluation_measures import (compute_per_intersection_macro_f1, compute_psds_from_operating_points,
                                                     compute_psds_from_scores, log_sedeval_metrics)
        from .postprocess import write_sed_scores
        self._test_state()
        save_dir = os.path.join(self.exp_dir, "metrics_test")
        if self.evaluation:
            for who, raw, post in (("student", self.test_scores_raw_buffer_student, self.test_scores_postprocessed_buffer_student),
                                   ("teacher", self.test_scores_raw_buffer_teacher, self.test_scores_postprocessed_buffer_teacher)):
                write_sed_scores(raw, os.path.join(save_dir, f"{who}_scores", "raw"))
                write_sed_scores(post, os.path.join(save_dir, f"{who}_scores", "postprocessed"))
                print(f"\nRaw and postprocessed scores for {who} saved in: {os.path.join(save_dir, who + '_scores')}")
            results = {}
        else:
            data = self.hparams["data"]
            results = {}
            ground_truth, audio_durations = self._scored_ground_truth(data["test_tsv"], data["test_dur"],
                                                                      self.test_scores_postprocessed_buffer_student)
            for who, buf, buf05, post in (
                    ("student", self.test_psds_buffer_student, self.decoded_student_05_buffer, self.test_scores_postprocessed_buffer_student),
                    ("teacher", self.test_psds_buffer_teacher, self.decoded_teacher_05_buffer, self.test_scores_postprocessed_buffer_teacher)):
                results[f"test/{who}/psds1_psds_eval"] = compute_psds_from_operating_points(
                    buf, data["test_tsv"], data["test_dur"], dtc_threshold=0.7, gtc_threshold=0.7, alpha_ct=0, alpha_st=1,
                    save_dir=os.path.join(save_dir, who, "scenario1"))
                results[f"test/{who}/psds1_sed_scores_eval"] = compute_psds_from_scores(
                    post, ground_truth, audio_durations, dtc_threshold=0.7, gtc_threshold=0.7, cttc_threshold=None, alpha_ct=0,
                    alpha_st=1, save_dir=os.path.join(save_dir, who, "scenario1"))
                results[f"test/{who}/psds2_psds_eval"] = compute_psds_from_operating_points(
                    buf, data["test_tsv"], data["test_dur"], dtc_threshold=0.1, gtc_threshold=0.1, cttc_threshold=0.3,
                    alpha_ct=0.5, alpha_st=1, save_dir=os.path.join(save_dir, who, "scenario2"))
                results[f"test/{who}/psds2_sed_scores_eval"] = compute_psds_from_scores(
                    post, ground_truth, audio_durations, dtc_threshold=0.1, gtc_threshold=0.1, cttc_threshold=0.3, alpha_ct=0.5,
                    alpha_st=1, save_dir=os.path.join(save_dir, who, "scenario2"))
                results[f"test/{who}/event_f1_macro"] = log_sedeval_metrics(buf05, data["test_tsv"], os.path.join(save_dir, who))[0]
                results[f"test/{who}/intersection_f1_macro"] = compute_per_intersection_macro_f1(
                    {"0.5": buf05}, data["test_tsv"], data["test_dur"])
            results["hp_metric"] = torch.tensor(max(results["test/student/psds1_psds_eval"], results["test/student/psds2_psds_eval"]))
        logger = getattr(self, "logger", None)
        if logger is not None:
            logger.log_metrics(results)
            logger.log_hyperparams(self.hparams, results)
        for key in results.keys():
            self.log(key, results[key], prog_bar=True, logger=True)
        return results
