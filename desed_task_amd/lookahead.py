"""`train_dataloader()`'s loader with a one-batch look-ahead (recipes/dcase2023_task4_baseline/local/sed_trainer.py:913-920 returns a
plain `torch.utils.data.DataLoader`; this IS one -- same constructor, same batches in the same order, `isinstance(..., DataLoader)`
holds for Lightning's checks).

Why: the software-pipelined step (SEDTask4.launch_prefetch) runs the front half of batch k + 1 -- mel, mixup, log / min-max -- and the
teacher's CNN forward on a side stream under step k's backward, so step k must know batch k + 1.  A hand-written loop passes it
(`StepDriver.run_step(batch, i, next_batch=...)`); under `pl.Trainer.fit` nothing does -- the trainer only ever hands
`training_step` ONE batch.  The loader is the part of the LightningModule surface that sees the stream of batches, so it keeps the
batches it has handed out, in order, until the module releases them, and fetches one more than it was asked for:

    find(batch)        -> key of a batch this loader yielded (identity of the yielded object, or of its first tensor), else None
    batch_after(key)   -> the batch that follows it in this epoch (already yielded to a prefetching trainer, or fetched now), or None
                          at the end of the epoch
    release(key)       -> forget everything up to and including `key`

Nothing else changes: an epoch yields exactly the batches the plain DataLoader would, the worker processes and the sampler are torch's.
(SEDTask4 asks two batches ahead: k + 1 for the step's side branch, k + 2 to upload it while step k runs.  With `num_workers = 0` a data
set that draws random numbers in `__getitem__` -- the reference's random crop of over-long clips -- therefore draws them up to two
batches earlier relative to the step's own draws than under a plain loop; with worker processes, the recipes' setting, nothing moves.)
"""
import collections

import torch
from torch.utils.data import DataLoader


def first_tensor(batch):
    if torch.is_tensor(batch):
        return batch
    if isinstance(batch, (list, tuple)):
        for b in batch:
            t = first_tensor(b)
            if t is not None:
                return t
    return None


class _LookaheadIter:
    def __init__(self, loader, it, epoch):
        self.loader, self.it, self.epoch = loader, it, epoch
        self.window = collections.deque()      # [seq, batch] fetched and not yet released, oldest first
        self.fetched = 0                        # batches pulled from the underlying iterator so far
        self.handed = 0                         # batches returned by __next__ so far
        self.exhausted = False

    def __iter__(self):
        return self

    def _fetch(self):
        if self.exhausted:
            return False
        try:
            b = next(self.it)
        except StopIteration:
            self.exhausted = True
            return False
        self.window.append([self.fetched, b])
        self.fetched += 1
        while len(self.window) > self.loader.MAX_HELD:      # a loop that never releases (hooks mode): bounded memory
            self.window.popleft()
        return True

    def at(self, seq):
        while self.fetched <= seq:
            if not self._fetch():
                return None
        for s, b in self.window:
            if s == seq:
                return b
        return None

    def __next__(self):
        b = self.at(self.handed)
        if b is None:
            raise StopIteration
        self.handed += 1
        self.at(self.handed)            # stay one ahead (the workers of a DataLoader have it ready anyway)
        return b


class LookaheadLoader(DataLoader):
    MAX_HELD = 6            # a trainer that prefetches one batch holds two; the look-ahead adds two (next batch + the one being uploaded)

    _cur = None
    _epochs = 0

    def __iter__(self):
        self._epochs += 1
        self._cur = _LookaheadIter(self, super().__iter__(), self._epochs)
        return self._cur

    def find(self, batch):
        """(epoch, seq) of `batch` if the CURRENT iteration yielded it and has not released it yet."""
        cur = self._cur
        if cur is None:
            return None
        t = first_tensor(batch)
        for s, b in cur.window:
            if s < cur.handed and (b is batch or (t is not None and first_tensor(b) is t)):
                return (cur.epoch, s)
        return None

    def batch_after(self, key):
        cur = self._cur
        if cur is None or key is None or key[0] != cur.epoch:
            return None
        return cur.at(key[1] + 1)

    def release(self, key):
        cur = self._cur
        if cur is None or key is None or key[0] != cur.epoch:
            return
        while cur.window and cur.window[0][0] <= key[1]:
            cur.window.popleft()


class BatchList(torch.utils.data.Dataset):
    """A data set whose items are whole, ready batches (e.g. synthetic clips resident in HBM): `train_dataloader()` then iterates it
    with `batch_size=None` -- no sampler, no collation."""
    yields_batches = True

    def __init__(self, batches):
        self.batches = list(batches)

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        return self.batches[i]
