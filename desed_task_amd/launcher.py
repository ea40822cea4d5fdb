"""One-process-per-GPU driver of the mean-teacher step (the reference cannot start more than one GPU:
recipes/dcase2023_task4_baseline/train_sed.py:269-276).

`run_step` reproduces Lightning 1.9's automatic-optimisation order for one batch:
    training_step -> on_before_zero_grad (EMA) -> zero_grad -> backward -> [grad all-reduce] -> optimizer.step
    -> scheduler.step
with two MI355X-side differences: the EMA kernel runs on a side HIP stream concurrently with backward (it only
reads the student parameters that backward also only reads; Adam waits for it), and under data parallelism the
flat gradient arena is averaged with ONE RCCL all-reduce over xGMI (1,112,420 floats; the 1/world factor is
folded into the Adam kernel).  BN statistics and mixup stay rank-local, like the single-GPU reference.

State that diverges across ranks: the student's and the teacher's BatchNorm running statistics (each rank
normalises its own clips; momentum 0.99 makes them ~ the last batch).  Policy: `average_bn_buffers(task)` -- ONE all-reduce of
the 2 x 1 248 floats -- before every validation / test pass and before `save_checkpoint`, which writes from rank 0 only.
`RankShardedBatchSampler` gives rank r the batches r, r + world, ... of the recipe's ConcatDatasetBatchSampler.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()       # (several ranks on one GPU only make sense with the gloo test backend)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SED_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


class StepDriver:
    def __init__(self, task, world_size=1, ema_side_stream=True):
        self.task = task
        self.world = world_size
        self.opt = task.opt
        self.sched = task.scheduler["scheduler"]
        self.arena = getattr(task.sed_student, "arena", None)
        dev = next(task.sed_student.parameters()).device
        self.side = torch.cuda.Stream(device=dev) if (ema_side_stream and dev.type == "cuda") else None
        if hasattr(self.opt, "grad_scale"):
            self.opt.grad_scale = 1.0 / world_size

    def allreduce_grads(self):
        if self.world <= 1:
            return
        arena = self.arena
        if arena is not None:
            flat = arena.gather_grads()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if not hasattr(self.opt, "grad_scale"):
                flat.div_(self.world)
            if not arena.grads_are_flat():        # gradients lived elsewhere: scatter the averaged values back
                with torch.no_grad():
                    for p, o in zip(arena.params, arena.offsets):
                        if p.grad is not None:
                            p.grad.copy_(flat[o:o + p.numel()].view(p.shape))
        else:
            for p in self.task.sed_student.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    p.grad.div_(self.world)

    def run_step(self, batch, batch_idx=0):
        task = self.task
        loss = task.training_step(batch, batch_idx)
        if self.side is not None:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)                       # teacher forward has finished reading theta_t
            with torch.cuda.stream(self.side):
                task.on_before_zero_grad()
        else:
            task.on_before_zero_grad()
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.allreduce_grads()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)   # Adam overwrites theta_s that the EMA reads
        self.opt.step()
        task.lr_scheduler_step(self.sched, 0, None)
        return loss


def bn_buffers(task):
    """Running mean / variance tensors of the student and the teacher, in a fixed order."""
    out = []
    for model in (task.sed_student, task.sed_teacher):
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                out += [m.running_mean, m.running_var]
    return out


def average_bn_buffers(task, world_size=None):
    """Replace every BN running statistic by its mean over the ranks (one all-reduce of one flat tensor).  No-op at world 1."""
    world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
    if world <= 1:
        return
    bufs = bn_buffers(task)
    flat = torch.cat([b.detach().reshape(-1) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()


def save_checkpoint(task, path, world_size=None):
    """Averages the BN buffers, then rank 0 writes {"sed_student", "sed_teacher"} state dicts (on_save_checkpoint layout,
    sed_trainer.py:603-606); every rank returns after the file exists."""
    average_bn_buffers(task, world_size)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == 0:
        torch.save(task.on_save_checkpoint({}), path)
    if dist.is_initialized():
        dist.barrier()


class RankShardedBatchSampler:
    """Rank-strided view of a batch sampler (desed_task/dataio/sampler.py:69-80 yields whole [synth | weak | unlabelled]
    batches): rank r iterates batches r, r + world, ...; all ranks get the same number of batches (the tail is dropped)."""

    def __init__(self, batch_sampler, rank, world_size):
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world_size

    def set_epoch(self, epoch):
        if hasattr(self.batch_sampler, "set_epoch"):
            self.batch_sampler.set_epoch(epoch)

    def __len__(self):
        return len(self.batch_sampler) // self.world

    def __iter__(self):
        n = len(self) * self.world
        for i, batch in enumerate(self.batch_sampler):
            if i >= n:
                break
            if i % self.world == self.rank:
                yield batch
