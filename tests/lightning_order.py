"""A stand-in for `pl.Trainer.fit` (recipes/dcase2023_task4_baseline/train_sed.py:278-299): Lightning is not installed in this image, so
this loop calls the LightningModule hooks -- and ONLY hooks of the module, plus `optimizer.step(closure)` -- in the order
pytorch_lightning 1.9's automatic optimisation does (loops/fit_loop.py, loops/epoch/training_epoch_loop.py,
loops/optimization/optimizer_loop.py):

    configure_optimizers(); train_dataloader()                                   once per fit
    per epoch:    model.train(); on_train_epoch_start()
      per batch:  transfer_batch_to_device(batch, device, 0); on_train_batch_start(batch, i)
                  optimizer_step(epoch, i, optimizer, 0, closure)                -> optimizer.step(closure=closure)
                      closure:  training_step(batch, i)  ->  loss / accumulate_grad_batches
                                on_before_zero_grad(optimizer); optimizer_zero_grad(epoch, i, optimizer, 0)
                                backward(loss, optimizer, 0)
                                on_before_optimizer_step(optimizer, 0)           (+ gradient clipping: gradient_clip 0. -> nothing)
                  lr_scheduler_step(scheduler, 0, None)                          ("interval": "step")
                  on_train_batch_end(out, batch, i)
                  [validation every `check_val_every_n_epoch` epochs: model.eval(); validation_step ...; validation_epoch_end]
    on_train_end()

Test / bench infrastructure: it holds no training logic of its own -- whatever the module does inside its hooks is the product.
"""
import torch


def _hook(module, name, *args):
    fn = getattr(module, name, None)
    return fn(*args) if callable(fn) else None


class Trainer:
    """The arguments of train_sed.py:278-296 that change what the loop calls; everything else is accepted and ignored."""

    def __init__(self, max_epochs=1, limit_train_batches=1.0, limit_val_batches=0, accumulate_grad_batches=1, device=None,
                 check_val_every_n_epoch=1, on_step=None, abandon=None, **ignored):
        self.max_epochs, self.limit_train, self.limit_val = max_epochs, limit_train_batches, limit_val_batches
        self.accumulate, self.device, self.val_every, self.on_step = accumulate_grad_batches, device, check_val_every_n_epoch, on_step
        self.abandon = abandon or {}     # {epoch: n}: leave that epoch's loop after n batches WITHOUT the module knowing beforehand (an
        self.current_epoch = self.global_step = 0     # interrupted epoch / `max_steps`: `num_training_batches` stays what it was)
        self.num_training_batches = None
        self.losses = []

    def _limit(self, n, limit):
        return int(n * limit) if isinstance(limit, float) else min(n, int(limit))

    def fit(self, model, ckpt_path=None):
        optimizers, schedulers = model.configure_optimizers()
        opt, sched = optimizers[0], schedulers[0]
        assert sched.get("interval", "epoch") == "step"
        device = self.device or next(model.parameters()).device
        model.trainer = self
        loader = model.train_dataloader()
        self.num_training_batches = self._limit(len(loader), self.limit_train)
        _hook(model, "on_train_start")
        for epoch in range(self.max_epochs):
            self.current_epoch = model.current_epoch = epoch
            model.train()
            _hook(model, "on_train_epoch_start")
            for i, batch in enumerate(loader):
                if i >= self.num_training_batches or i >= self.abandon.get(epoch, i + 1):
                    break
                batch = model.transfer_batch_to_device(batch, device, 0)
                _hook(model, "on_train_batch_start", batch, i)
                out = {}

                def closure():
                    loss = model.training_step(batch, i)
                    out["loss"] = closure_loss = loss / self.accumulate          # (closure.py: a new tensor, never in place)
                    model.on_before_zero_grad(opt)
                    model.optimizer_zero_grad(epoch, i, opt, 0)
                    model.backward(closure_loss, opt, 0)
                    _hook(model, "on_before_optimizer_step", opt, 0)
                    return closure_loss

                model.optimizer_step(epoch, i, opt, 0, closure)
                self.global_step += 1
                model.lr_scheduler_step(sched["scheduler"], 0, None)
                _hook(model, "on_train_batch_end", out, batch, i)
                self.losses.append(out["loss"].detach())
                if self.on_step is not None:
                    self.on_step(self, model, i)
            _hook(model, "on_train_epoch_end")
            if self.limit_val and (epoch + 1) % self.val_every == 0:
                model.eval()
                with torch.no_grad():
                    vl = model.val_dataloader()
                    for i, batch in enumerate(vl):
                        if i >= self._limit(len(vl), self.limit_val):
                            break
                        model.validation_step(model.transfer_batch_to_device(batch, device, 0), i)
                    model.validation_epoch_end([])
        _hook(model, "on_train_end")
        return self
