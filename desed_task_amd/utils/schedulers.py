"""Step-wise schedulers with the reference's interface (desed_task/utils/schedulers.py:8-104): the same object
scales the learning rate and the mean-teacher consistency weight (sed_trainer.py:329-332).  Host-side scalars."""
import numpy as np


class BaseScheduler(object):
    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.step_num = 0

    def _get_lr(self):
        raise NotImplementedError

    def _set_lr(self, lr):
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def step(self, metrics=None, epoch=None):
        self.step_num += 1
        self._set_lr(self._get_lr())

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}


class ExponentialWarmup(BaseScheduler):
    """lr = max_lr * exp(exponent * (1 - min(step, R)/R)^2), optional cosine annealing; step_num starts at 1."""

    def __init__(self, optimizer, max_lr, rampup_length, exponent=-5.0, start_annealing=None, max_steps=None, min_lr=1e-8):
        super().__init__(optimizer)
        self.rampup_len = rampup_length
        self.max_lr = max_lr
        self.step_num = 1
        self.exponent = exponent
        self.start_annealing = start_annealing
        self.max_steps = max_steps
        self.min_lr = min_lr

    def _rampup(self):
        current = np.clip(self.step_num, 0.0, self.rampup_len)
        phase = 1.0 - current / self.rampup_len
        return float(np.exp(self.exponent * phase * phase))

    def _get_scaling_factor(self):
        if self.rampup_len == 0:
            return 1.0
        if self.start_annealing is not None and self.step_num >= self.start_annealing:
            done = self.step_num - self.start_annealing
            span = self.max_steps - self.start_annealing
            return max(self.min_lr / self.max_lr, np.cos(done * np.pi / (2 * span)))
        return self._rampup()

    def _get_lr(self):
        return self.max_lr * self._get_scaling_factor()
