"""One-process-per-GPU driver of the mean-teacher step (the reference cannot start more than one GPU:
recipes/dcase2023_task4_baseline/train_sed.py:269-276).

`run_step` reproduces Lightning 1.9's automatic-optimisation order for one batch:
    training_step -> on_before_zero_grad (EMA) -> zero_grad -> backward -> [grad all-reduce] -> optimizer.step
    -> scheduler.step
with two MI355X-side differences: the EMA kernel runs on a side HIP stream concurrently with backward (it only
reads the student parameters that backward also only reads; Adam waits for it), and under data parallelism the
flat gradient arena is summed over the ranks in two buckets (heads + BiGRU launched under the CNN backward, then the
CNN) with RCCL all-reduces over xGMI (1,112,420 floats in total; the 1/world factor is folded into the Adam kernel).  BN statistics and mixup stay rank-local, like the single-GPU reference.

State that diverges across ranks: the student's and the teacher's BatchNorm running statistics (each rank
normalises its own clips; momentum 0.99 makes them ~ the last batch).  Policy: `average_bn_buffers(task)` -- ONE all-reduce of
the 2 x 1 248 floats -- before every validation / test pass and before `save_checkpoint`, which writes from rank 0 only.
`RankShardedBatchSampler` gives rank r the batches r, r + world, ... of the recipe's ConcatDatasetBatchSampler.
"""
import os

import torch
import torch.distributed as dist

from . import ops as _ops


# Enqueue order of the pipelined side branch (SEDTask4.launch_prefetch) relative to backward.  The dependencies are the same either
# way -- the branch forks before backward and joins after it --, but a hipGraph replay keeps the FIRST-captured successor of a fork on
# the main hardware queue and moves the other one to a second queue, which on this runtime only started ~0.6 ms later: with the side
# branch captured first it was the backward pass that waited.  Captured after backward, the side branch is the one that moves.
PREFETCH_ENQUEUE_LATE = os.environ.get("SED_PF_LATE", "1") != "0"


class ExchangeProbe:
    """Where the time of a data-parallel step's TAIL goes (bench.py's `dist` object): marks on the compute stream
    (HIP events, so device time) and on the host clock at
        step_start -> backward_done -> [A_issued -> A_done ->] exchange_done -> adam_done
    `backward_done -> exchange_done` on the DEVICE clock is the exposed cost of the gradient exchange (the collective(s) plus
    whatever the host added by enqueueing late); the same interval on the HOST clock is what the host spent issuing it -- if that
    exceeds the device interval the step is host-bound there (the GPU idles between the graph and Adam).  Collectives are issued
    through torch.distributed on the backend's own stream; Work.wait() makes the compute stream wait for them, so a mark recorded
    right after the wait lies behind the collective on the device timeline.  On a CPU device only the host clock exists."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.steps, self.cur = [], None

    def begin(self):
        self.cur = []
        self.mark("step_start")

    def mark(self, tag):
        if self.cur is None:
            return
        import time
        ev = None
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        self.cur.append((tag, ev, time.perf_counter()))

    def end(self):
        if self.cur is not None:
            self.steps.append(self.cur)
        self.cur = None

    def summary(self):
        """{"<tag a> -> <tag b>": {"device_us": median, "host_us": median}, ...} over the recorded steps + the exposed totals."""
        import statistics
        if self.cuda:
            torch.cuda.synchronize()
        seg = {}
        for marks in self.steps:
            for (ta, ea, ha), (tb, eb, hb) in zip(marks, marks[1:]):
                d = seg.setdefault("%s -> %s" % (ta, tb), {"device_us": [], "host_us": []})
                d["host_us"].append((hb - ha) * 1e6)
                if ea is not None:
                    d["device_us"].append(ea.elapsed_time(eb) * 1e3)
            tags = [m[0] for m in marks]
            if "backward_done" in tags and "exchange_done" in tags:
                a, b = marks[tags.index("backward_done")], marks[tags.index("exchange_done")]
                d = seg.setdefault("exposed_exchange (backward_done -> exchange_done)", {"device_us": [], "host_us": []})
                d["host_us"].append((b[2] - a[2]) * 1e6)
                if a[1] is not None:
                    d["device_us"].append(a[1].elapsed_time(b[1]) * 1e3)
        out = {}
        for k, d in seg.items():
            out[k] = {"device_us": round(statistics.median(d["device_us"]), 1) if d["device_us"] else None,
                      "host_us": round(statistics.median(d["host_us"]), 1), "samples": len(d["host_us"])}
        return out


def rehearsing():
    """SED_DDP_REHEARSE=1: run the data-parallel step structure (graph split, bucketed all-reduces, eager Adam, start-up broadcast)
    on a process group of ONE rank.  The sums over one rank change no bit, so the step must equal the plain single-GPU step exactly
    -- which is what tests/test_gpu_ddp_graph.py checks over RCCL on the 1-GPU boxes, and what `bench.py --rehearse-exchange` times
    (the per-step cost of the exchange machinery without any link time)."""
    return os.environ.get("SED_DDP_REHEARSE") == "1"


def init_distributed(backend=None):
    """Read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()       # (several ranks on one GPU only make sense with the gloo test backend)
    if (world > 1 or rehearsing()) and not dist.is_initialized():
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
    """One mean-teacher step in Lightning 1.9's order, optionally data-parallel.

    Gradient exchange (world_size > 1), as BASELINE.json's north_star states it: the flat gradient arena is exchanged in TWO
    buckets that follow the order in which backward produces them --
      bucket A = everything behind the CNN in parameters() order (BiGRU, the two dense heads, cat_tf): complete as soon as the
                 recurrent stage's backward has run.  The student's autograd graph is cut at the CNN output, loss.backward()
                 therefore returns right there, the all-reduce of A is handed to RCCL as an asynchronous collective (its own
                 stream) and runs UNDER the CNN backward, which is enqueued next;
      bucket B = the CNN parameters, reduced after backward.
    Adam waits for both (and for the EMA side stream).  The sum / world_size average is folded into the Adam kernel.
    `overlap_allreduce=False` gives the single blocking all-reduce over the whole arena (same result, bit for bit: a sum over
    ranks per element either way)."""

    def __init__(self, task, world_size=1, ema_side_stream=True, overlap_allreduce=None, broadcast_init=True, gru_dw_side=True,
                 prefetch=None):
        """prefetch: None (the task's setting, default off) | "tails" | "backward" | "teacher" -- software-pipelined front half:
        run_step(batch, i, next_batch=...) announces the next batch; "tails" / "backward": its mel kernel runs on a side stream under
        this step's BiGRU phases (fork before the tails / before backward); "teacher": its whole front half (mel, mixup, log /
        min-max) AND the teacher's CNN forward run under this step's backward (SEDTask4.launch_prefetch).  The next run_step must
        be given exactly the announced batch."""
        self.task = task
        _ops.reset_loss_work(next(task.sed_student.parameters()).device)
        if prefetch is not None:
            task.prefetch_point = None if prefetch in ("off", False) else ("backward" if prefetch == "teacher" else prefetch)
            task.prefetch_level = "teacher" if prefetch == "teacher" else "features"
        self._announced = None
        self.check_announced = True     # False: the caller vouches for the order of the batches (SEDTask4's whole-step mode: loader keys)
        self.world = world_size
        # the gradient exchange runs at world > 1 -- and on a one-rank process group when rehearsing (see rehearsing())
        self.exchange = world_size > 1 or (rehearsing() and dist.is_initialized())
        self.opt = task.opt
        self.sched = task.scheduler["scheduler"]
        self.arena = getattr(task.sed_student, "arena", None)
        dev = next(task.sed_student.parameters()).device
        self.side = torch.cuda.Stream(device=dev) if (ema_side_stream and dev.type == "cuda") else None
        self._gru_dw_side_arg = bool(gru_dw_side)
        if hasattr(self.opt, "grad_scale"):
            self.opt.grad_scale = 1.0 / world_size
        if overlap_allreduce is None:
            # Default: bucket A under the CNN backward.  Exception: with the front half of the next step and the teacher's CNN
            # forward pipelined under this step's backward (prefetch "teacher") the step stays ONE graph -- the side branch must
            # join before the graph that forked it ends, and cutting backward in two would leave it the BiGRU part only -- and the
            # gradients go out as one all-reduce after backward (4.45 MB, latency-bound: tens of us against the ~0.2 ms the
            # pipelined branch saves).  SED_DDP_OVERLAP=1 / 0 forces either scheme (A/B on a real node).
            env = os.environ.get("SED_DDP_OVERLAP")
            overlap_allreduce = (env != "0") if env is not None else getattr(task, "prefetch_level", "features") != "teacher"
        self.overlap = bool(overlap_allreduce) and self.exchange and self.arena is not None
        # see ops.GRU_DW_SIDE / CNN_DW_SIDE: the weight-gradient GEMMs beside the backward chain, on only inside this driver's backward().
        # Under a gradient exchange (round 5): on when the gradients go out as ONE all-reduce after backward_joined() has joined the side
        # stream (the default of the pipelined step), over RCCL -- one-rank rehearsal 3.05 vs 3.22 ms, strict-equal to the plain step;
        # off with the bucketed overlap (bucket A is issued from inside the backward pass: a side section would still be running --
        # the rehearsal test fails with it) and over gloo (two ranks sharing ONE GPU, the only multi-rank configuration this container
        # can run, went 20 x slower with the side-stream launches, unexplained).  SED_GRU_DW_SIDE=1 / 0 forces it (A/B on a real node).
        env_side = os.environ.get("SED_GRU_DW_SIDE")
        side_ok = not self.exchange
        if self.exchange and not self.overlap and dist.is_initialized():
            side_ok = dist.get_backend() == "nccl"
        if env_side is not None and self.exchange:
            side_ok = env_side == "1"
        self.gru_dw_side = self._gru_dw_side_arg and (dev.type == "cuda" or _ops.SIDE_ON_CPU) and side_ok
        self.bucket_log = []            # [(tag, first float, number of floats)] of the collectives of the last step (tests)
        self._work_a = None
        self.probe = None               # an ExchangeProbe while bench.py times the tail of the step
        if self.exchange and broadcast_init and dist.is_initialized():
            self.broadcast_state()

    # ---- start-up: every rank continues from rank 0's weights and BatchNorm buffers --------------------------------
    def broadcast_state(self):
        """Ranks must not depend on having been seeded alike: parameters (student and teacher) and BN buffers come from rank 0."""
        some = None
        for model in (self.task.sed_student, self.task.sed_teacher):
            arena = getattr(model, "arena", None)
            if arena is not None:
                _gloo_fence(arena.flat)
                dist.broadcast(arena.flat, src=0)
                some = arena.flat
            else:
                for p in model.parameters():
                    dist.broadcast(p.data, src=0)
                    some = p.data
        bufs = bn_buffers(self.task)
        if bufs:
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            _gloo_fence(flat)
            dist.broadcast(flat, src=0)
            _gloo_fence(flat)
            off = 0
            with torch.no_grad():
                for b_ in bufs:
                    b_.copy_(flat[off:off + b_.numel()].view_as(b_))
                    off += b_.numel()
        if some is not None:
            _gloo_fence(some)       # (gloo writes its result back on streams of its own -- also on the source rank)

    # ---- gradient buckets -------------------------------------------------------------------------------------
    def bucket_bounds(self):
        """(split, numel): bucket B = arena floats [0, split) (the CNN, first in parameters() order), bucket A = [split, numel)."""
        arena = self.task.sed_student.arena
        cnn = getattr(self.task.sed_student, "cnn", None)
        n_cnn = len(list(cnn.parameters())) if cnn is not None else 0
        split = arena.offsets[n_cnn] if n_cnn < len(arena.offsets) else arena.numel
        return split, arena.numel

    def _reduce_bucket(self, tag, lo, hi, async_op):
        flat = self.task.sed_student.arena.flat_grad
        self.bucket_log.append((tag, lo, hi - lo))
        if hi <= lo:
            return None
        _gloo_fence(flat)
        return dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=async_op)

    def backward(self, loss):
        """loss.backward() with the gradient exchange.  Overlap mode: the student's autograd graph is cut at the CNN output
        (CRNN.split_backward), so loss.backward() ends after the recurrent stage; bucket A is handed to RCCL there as an
        asynchronous collective and the CNN backward is enqueued behind it on the compute stream."""
        student = self.task.sed_student
        if not (self.overlap and getattr(student, "_cnn_boundary", None) is not None):
            self.backward_joined(loss)
            if hasattr(student, "backward_cnn"):
                student.backward_cnn()
            self._mark("backward_done")
            self.allreduce_grads()
            return
        self.backward_joined(loss)                       # bucket A holds the BiGRU weight gradients
        self._mark("backward_done")                      # (of the heads + BiGRU: bucket A is complete)
        self.launch_bucket_a()
        student.backward_cnn()
        self.finish_buckets()

    def backward_joined(self, loss):
        """loss.backward() with the BiGRU weight-gradient GEMMs on the side stream (ops.GRU_DW_SIDE), joined before returning."""
        prev, _ops.GRU_DW_SIDE = _ops.GRU_DW_SIDE, self.gru_dw_side and _ops.GRU_DW_SIDE_ALLOWED
        prev_c, _ops.CNN_DW_SIDE_NOW = _ops.CNN_DW_SIDE_NOW, not (self.exchange and self.overlap)     # (bucket A goes out from inside backward)
        del _ops._deferred[:]               # (nothing may be left over from a backward pass that raised)
        try:
            torch.autograd.backward(loss, _ops.unit_grad(loss.device))       # = loss.backward() without the ones_like fill
        finally:
            _ops.GRU_DW_SIDE = prev
            _ops.CNN_DW_SIDE_NOW = prev_c
        _ops.join_side_stream(loss.device)  # also launches what ops.defer_off_chain() still holds

    def launch_bucket_a(self):
        arena = self.task.sed_student.arena
        split, n = self.bucket_bounds()
        base = arena.flat_grad.data_ptr()
        flat_a = all(p.grad is None or p.grad.data_ptr() == base + 4 * o
                     for p, o in zip(arena.params, arena.offsets) if o >= split)
        self._work_a = (self._reduce_bucket("A", split, n, async_op=True) or True) if flat_a else None
        self._mark("A_issued")

    def finish_buckets(self):
        arena = self.task.sed_student.arena
        split, n = self.bucket_bounds()
        if self._work_a is None or not arena.grads_are_flat():
            # some gradient lives outside the arena: the blocking whole-arena path (bucket A, if in flight, first completes and
            # is excluded so that nothing is summed twice)
            if self._work_a is not None and self._work_a is not True:
                self._work_a.wait()
            done_a = self._work_a is not None
            self._work_a = None
            flat = arena.gather_grads()
            self.bucket_log.append(("B" if done_a else "AB", 0, split if done_a else n))
            _gloo_fence(flat)
            dist.all_reduce(flat[:split] if done_a else flat, op=dist.ReduceOp.SUM)
            _gloo_fence(flat)
            self._finish_scale(arena)
            return
        self._mark("cnn_backward_enqueued")
        if self._work_a is not None and self._work_a is not True:
            self._work_a.wait()
        self._mark("A_done")
        wb = self._reduce_bucket("B", 0, split, async_op=True)
        if wb is not None:
            wb.wait()
        self._work_a = None
        # gloo only: its copy-back of the reduced buckets runs on streams of its own, and Work.wait() did not reliably order it in
        # front of a non-default compute stream on this stack -- the one two-rank run in nine that differed (2e-5) used this path
        _gloo_fence(arena.flat_grad)
        self._finish_scale(arena)

    def _finish_scale(self, arena):
        flat = arena.flat_grad
        if not hasattr(self.opt, "grad_scale"):
            flat.div_(self.world)
        if not arena.grads_are_flat():        # gradients lived elsewhere: scatter the averaged values back
            with torch.no_grad():
                for p, o in zip(arena.params, arena.offsets):
                    if p.grad is not None:
                        p.grad.copy_(flat[o:o + p.numel()].view(p.shape))

    def allreduce_grads(self):
        """The blocking exchange: ONE all-reduce over the whole flat gradient arena."""
        if not self.exchange:
            return
        arena = self.task.sed_student.arena if self.arena is not None else None
        if arena is not None:
            flat = arena.gather_grads()
            self.bucket_log.append(("AB", 0, arena.numel))
            _gloo_fence(flat)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            _gloo_fence(flat)
            self._finish_scale(arena)
        else:
            for p in self.task.sed_student.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    if not hasattr(self.opt, "grad_scale"):
                        p.grad.div_(self.world)

    def _mark(self, tag):
        if self.probe is not None:
            self.probe.mark(tag)

    def arm_overlap(self):
        """Cut the student's autograd graph at the CNN output for this step when the overlapped exchange is on."""
        self.bucket_log = []
        self._work_a = None
        student = self.task.sed_student
        if hasattr(student, "split_backward"):
            student.split_backward = bool(self.overlap)

    def announce(self, batch, next_batch, staged=None):
        """Pipelined front-end protocol: the batch of this step must be the one announced by the previous step (same storage and
        shape: its features are already in the task's feature buffer -- or the driver's staging buffer `staged` the announced waveforms
        were copied into); `next_batch` is announced for this step's prefetch."""
        task = self.task
        if getattr(task, "prefetch_point", None) is None:
            return
        if getattr(task, "_feat_ready", False) or (getattr(task, "_pro", None) or {}).get("ready"):
            key = (batch[0].data_ptr(), tuple(batch[0].shape))
            staged_key = (staged.data_ptr(), tuple(staged.shape)) if staged is not None else None
            if self.check_announced and self._announced is not None and key != self._announced and key != staged_key:
                raise RuntimeError("run_step got another batch than the one announced as next_batch by the previous step")
        nxt = next_batch[0] if next_batch is not None else None
        self._announced = (nxt.data_ptr(), tuple(nxt.shape)) if nxt is not None else None
        if getattr(task, "prefetch_level", "features") == "teacher" and nxt is not None:
            if len(next_batch) < 2 or next_batch[1] is None:
                raise ValueError('prefetch "teacher": next_batch must carry the labels too (they are mixed one step early)')
            extras = task.next_batch_extras(next_batch) if hasattr(task, "next_batch_extras") else {}
            task.set_next_batch(nxt, next_batch[1], extras)
        else:
            task.set_next_batch(nxt, None)

    def run_step(self, batch, batch_idx=0, next_batch=None, staged=None):
        task = self.task
        if self.probe is not None:
            self.probe.begin()
        try:
            return self._run_step(batch, batch_idx, next_batch, staged)
        finally:
            if self.probe is not None:
                self.probe.end()

    def training_step_and_ema(self, batch, batch_idx):
        """training_step(), then `on_before_zero_grad` (the EMA) -- on the side stream when there is one, together with whatever the
        forward pass parked for it (ops.AFTER_FORWARD: the loss sums, which feed the log and not the backward pass)."""
        task = self.task
        park = (self.side is not None or _ops.SIDE_ON_CPU) and _ops.PARK_LOSS_SUMS
        prev, _ops.AFTER_FORWARD = _ops.AFTER_FORWARD, ([] if park else None)
        inside, task._in_driver = getattr(task, "_in_driver", False), True      # (SEDTask4.training_step: the step BODY, not the surface)
        try:
            loss = task.training_step(batch, batch_idx)
        finally:
            parked, _ops.AFTER_FORWARD = _ops.AFTER_FORWARD, prev
            task._in_driver = inside
        if self.side is not None:
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)                       # teacher forward has finished reading theta_t
            if parked:
                _ops.run_after_forward(parked, self.side)
            with torch.cuda.stream(self.side):
                task.on_before_zero_grad()
        else:
            if parked:
                _ops.run_after_forward(parked, None)
            task.on_before_zero_grad()
        return loss

    def _run_step(self, batch, batch_idx, next_batch, staged):
        task = self.task
        self.announce(batch, next_batch, staged=staged)
        self.arm_overlap()
        loss = self.training_step_and_ema(batch, batch_idx)
        self.opt.zero_grad(set_to_none=True)
        late = PREFETCH_ENQUEUE_LATE and loss.is_cuda
        fork = None
        if late:                                # the side branch forks HERE but is enqueued after backward (see PREFETCH_ENQUEUE_LATE)
            fork = torch.cuda.Event()
            fork.record()
        elif hasattr(task, "launch_prefetch"):
            task.launch_prefetch("backward", after=(self.side,))     # (the teacher forward of the next step reads the EMA's result)
        self.backward(loss)
        if late and hasattr(task, "launch_prefetch"):
            task.launch_prefetch("backward", after=(self.side,), fork_event=fork)
        if hasattr(task, "join_prefetch"):
            task.join_prefetch()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)   # Adam overwrites theta_s that the EMA reads
        self._mark("exchange_done")
        self.opt.step()
        self._mark("adam_done")
        task.lr_scheduler_step(self.sched, 0, None)
        return loss


def _gloo_fence(t):
    """gloo stages GPU tensors through the host on streams of its own; on this ROCm stack its ordering against a NON-default current
    stream was not reliable (two test ranks sharing one GPU gave run-to-run different sums until the device was synchronised around
    the collective).  gloo is the CPU / test backend -- RCCL collectives are stream-ordered and take no fence."""
    if t.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo" and os.environ.get("SED_GLOO_FENCE", "1") != "0":
        torch.cuda.synchronize(t.device)


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


def checkpoint_dict(task, epoch=0):
    """A Lightning-1.9-shaped checkpoint of the run: what `train_sed.py` reads back -- `state_dict` (`sed_student.*` /
    `sed_teacher.*` keys, :302 and --test_from_checkpoint :367-374), `hyper_parameters`, `epoch` -- plus what a resume needs
    (`optimizer_states` in torch.optim.Adam's layout, `lr_schedulers` with `step_num`, `global_step`) and the
    `on_save_checkpoint` extras (sed_trainer.py:603-606)."""
    sched = task.scheduler["scheduler"] if task.scheduler is not None else None
    ckpt = {
        "epoch": int(epoch),
        "global_step": int(sched.step_num - 1) if sched is not None else 0,
        "pytorch-lightning_version": "1.9.0",
        "state_dict": {k: v.detach().cpu().clone() for k, v in task.state_dict().items()},
        "optimizer_states": [task.opt.state_dict()] if task.opt is not None else [],
        "lr_schedulers": [sched.state_dict()] if sched is not None else [],
        "hyper_parameters": dict(task.hparams),
        "dropout_rng_state": _ops.dropout_rng_state(),      # the private dropout-seed stream continues where it stopped on resume
    }
    task.on_save_checkpoint(ckpt)
    return ckpt


def save_checkpoint(task, path, world_size=None, epoch=0):
    """Averages the BN buffers over the ranks, then rank 0 writes `checkpoint_dict(task)`; every rank returns after the file
    exists."""
    average_bn_buffers(task, world_size)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == 0:
        torch.save(checkpoint_dict(task, epoch), path)
    if dist.is_initialized():
        dist.barrier()


def load_checkpoint(task, path_or_dict, resume=True):
    """Inverse of save_checkpoint on every rank: weights + BN buffers of both models; with resume=True also Adam's moments / step
    and the scheduler's step_num (which drives lr, the consistency ramp-up and the EMA factor).  Also accepts the reference's
    own Lightning checkpoints (same keys)."""
    ckpt = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu", weights_only=False)
    task.load_state_dict(ckpt["state_dict"])
    if hasattr(task, "reset_pipeline"):
        task.reset_pipeline()           # a front half prefetched with the previous weights must not be consumed
    if resume:
        if ckpt.get("optimizer_states") and task.opt is not None:
            task.opt.load_state_dict(ckpt["optimizer_states"][0])
        if ckpt.get("lr_schedulers") and task.scheduler is not None:
            task.scheduler["scheduler"].load_state_dict(ckpt["lr_schedulers"][0])
        if ckpt.get("dropout_rng_state") is not None:
            _ops.set_dropout_rng_state(ckpt["dropout_rng_state"])
    return ckpt


class RankShardedBatchSampler:
    """Rank-strided view of a batch sampler (desed_task/dataio/sampler.py:69-80 yields whole [synth | weak | unlabelled]
    batches): rank r iterates batches r, r + world, ...; all ranks get the same number of batches (the tail is dropped)."""

    def __init__(self, batch_sampler, rank, world_size, seed=0):
        if not 0 <= rank < world_size:
            raise ValueError("rank out of range")
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world_size
        self.seed, self.epoch = int(seed), 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)
        if hasattr(self.batch_sampler, "set_epoch"):
            self.batch_sampler.set_epoch(epoch)

    def __len__(self):
        return len(self.batch_sampler) // self.world

    def __iter__(self):
        # Every rank must walk the SAME sequence of batches and keep its own share.  The recipe's samplers draw from torch's
        # global CPU generator, whose state differs across ranks as soon as anything rank-local consumed it (mixup permutations,
        # a rank-dependent seed): the epoch's batches are therefore drawn under a forked generator seeded with (seed, epoch)
        # only, and the caller's generator state is left untouched: the CPU state is saved / restored by fork_rng, and only the
        # CPU generator is seeded (torch.manual_seed() would also reseed every GPU generator -- the SpecAugment draws of all ranks
        # would then coincide and the user's GPU seeding would be lost).
        n = len(self) * self.world
        with torch.random.fork_rng(devices=[]):
            torch.default_generator.manual_seed(self.seed + self.epoch)
            batches = []
            for i, batch in enumerate(self.batch_sampler):
                if i >= n:
                    break
                batches.append(list(batch))
        self.epoch += 1             # a loop that never calls set_epoch() still gets a new order every epoch (same on every rank)
        for i, batch in enumerate(batches):
            if i % self.world == self.rank:
                yield batch


# ---- the one-command data-parallel run (SURVEY 8e "Launcher") ------------------------------------------------------------------------
#     python -m desed_task_amd.launcher --conf_file confs/default.yaml --log_dir exp/run [--gpus N] [--strong_real]
#                                       [--resume_from_checkpoint last.ckpt] [--test_from_checkpoint best.ckpt] [--fast_dev_run]
# from the recipe directory (so that `local.*` and `desed_task.dataio` -- the reference's data sets, samplers and label encoder, which
# are outside the hot path -- import as they do for train_sed.py).  It is what recipes/dcase2023_task4_baseline/train_sed.py:53-306 does,
# minus the refusal of more than one GPU (:269-276) and minus Lightning: N processes, one per GPU (self-launched through
# torch.distributed.run when the torchrun environment is absent), each building the SAME data sets, the recipe's
# ConcatDatasetBatchSampler behind RankShardedBatchSampler, `CRNN(**config["net"])`, `torch.optim.Adam`, `ExponentialWarmup`,
# `SEDTask4`; epochs of hipGraph-replayed steps (graph.GraphedStepDriver; launcher.StepDriver on a CPU device) with the next batch
# announced and uploaded ahead; every `validation_interval` epochs the BatchNorm statistics are averaged over the ranks and rank 0
# validates (same metrics, same objective), keeps `last.ckpt` / the best checkpoint in Lightning's layout (what --test_from_checkpoint
# and the reference's own loaders read) and decides about early stopping; after the last epoch rank 0 tests the best weights.
def _seed_all(seed, rank):
    """pl.seed_everything(seed) on rank 0; the other ranks draw their mixup / dropout / SpecAugment from seed + rank (every rank trains
    on its own clips with its own augmentation; the data split and the epoch's batch order are seeded separately, rank-independent)."""
    import random
    import numpy as np
    random.seed(seed + rank)
    np.random.seed(seed + rank)
    torch.manual_seed(seed + rank)
    _ops.reseed_dropout()


def build_run(config, log_dir, rank=0, world=1, strong_real=False, fast_dev_run=False, test_only=False, evaluation=False):
    """train_sed.py:79-246 -> the SEDTask4 of this package with the reference's data objects (imported from the reference checkout on
    PYTHONPATH; nothing of them is re-implemented here).  The train sampler is the recipe's ConcatDatasetBatchSampler seen through
    RankShardedBatchSampler: rank r trains on batches r, r + world, ... of every epoch."""
    import pandas as pd
    try:
        from desed_task.dataio import ConcatDatasetBatchSampler
        from desed_task.dataio.datasets import StronglyAnnotatedSet, UnlabeledSet, WeakSet
        from desed_task.utils.encoder import ManyHotEncoder
        from local.classes_dict import classes_labels
    except ImportError as e:  # pragma: no cover
        raise SystemExit("desed_task_amd.launcher needs the reference's data pipeline on PYTHONPATH (run it from the recipe directory, "
                         "like train_sed.py; INTEGRATION.md): %s" % e)
    from .nnet.CRNN import CRNN
    from .sed_trainer import SEDTask4
    from .utils.schedulers import ExponentialWarmup
    config = dict(config)
    config["log_dir"] = log_dir
    data, tr = config["data"], config["training"]
    encoder = ManyHotEncoder(list(classes_labels.keys()), audio_len=data["audio_max_len"], frame_len=config["feats"]["n_filters"],
                             frame_hop=config["feats"]["hop_length"], net_pooling=data["net_subsample"], fs=data["fs"])
    if not evaluation:
        test_data = StronglyAnnotatedSet(data["test_folder"], pd.read_csv(data["test_tsv"], sep="\t"), encoder, return_filename=True,
                                         pad_to=data["audio_max_len"])
    else:
        test_data = UnlabeledSet(data["eval_folder"], encoder, pad_to=None, return_filename=True)
    student = CRNN(**config["net"])
    if test_only:
        return SEDTask4(config, encoder=encoder, sed_student=student, test_data=test_data, fast_dev_run=fast_dev_run, evaluation=evaluation)
    synth = StronglyAnnotatedSet(data["synth_folder"], pd.read_csv(data["synth_tsv"], sep="\t"), encoder, pad_to=data["audio_max_len"])
    weak_df = pd.read_csv(data["weak_tsv"], sep="\t")
    train_weak_df = weak_df.sample(frac=tr["weak_split"], random_state=tr["seed"])
    valid_weak_df = weak_df.drop(train_weak_df.index).reset_index(drop=True)
    weak = WeakSet(data["weak_folder"], train_weak_df.reset_index(drop=True), encoder, pad_to=data["audio_max_len"])
    unlabeled = UnlabeledSet(data["unlabeled_folder"], encoder, pad_to=data["audio_max_len"])
    synth_val = StronglyAnnotatedSet(data["synth_val_folder"], pd.read_csv(data["synth_val_tsv"], sep="\t"), encoder, return_filename=True,
                                     pad_to=data["audio_max_len"])
    weak_val = WeakSet(data["weak_folder"], valid_weak_df, encoder, pad_to=data["audio_max_len"], return_filename=True)
    if strong_real:
        strong = StronglyAnnotatedSet(data["strong_folder"], pd.read_csv(data["strong_tsv"], sep="\t"), encoder, pad_to=data["audio_max_len"])
        parts = [torch.utils.data.ConcatDataset([strong, synth]), weak, unlabeled]
    else:
        parts = [synth, weak, unlabeled]
    sampler = RankShardedBatchSampler(ConcatDatasetBatchSampler([torch.utils.data.RandomSampler(x) for x in parts], tr["batch_size"]),
                                      rank, world, seed=int(tr["seed"] or 0))
    # The schedule counts optimizer steps: an epoch of N ranks has 1/N of the steps (N times the global batch), so the ramp of
    # `n_epochs_warmup` epochs is that many steps shorter -- the reference's formula with the per-rank epoch length
    steps_per_epoch = min(len(p) // (b * tr["accumulate_batches"]) for p, b in zip(parts, tr["batch_size"])) // world
    opt = torch.optim.Adam(student.parameters(), config["opt"]["lr"], betas=(0.9, 0.999))
    scheduler = {"scheduler": ExponentialWarmup(opt, config["opt"]["lr"], tr["n_epochs_warmup"] * max(steps_per_epoch, 1)), "interval": "step"}
    return SEDTask4(config, encoder=encoder, sed_student=student, opt=opt, train_data=torch.utils.data.ConcatDataset(parts),
                    valid_data=torch.utils.data.ConcatDataset([synth_val, weak_val]), test_data=test_data, train_sampler=sampler,
                    scheduler=scheduler, fast_dev_run=fast_dev_run, evaluation=evaluation)


def _run_eval(task, loader, device, step, limit):
    from ._lightning_standin import move_to_device
    task.eval()
    n = len(loader)
    n = n if limit is None else (int(n * limit) if isinstance(limit, float) else min(n, int(limit)))
    with torch.no_grad():
        for i, batch in enumerate(loader):
            if i >= n:
                break
            step(move_to_device(batch, device), i)


def fit(task, device, n_epochs, log_dir, world=1, rank=0, start_epoch=0, limit_train_batches=None, limit_val_batches=None,
        use_graph=None, best=None, log=None):
    """Epochs of data-parallel steps + validation on averaged BatchNorm statistics + checkpoints (what `trainer.fit` does for
    train_sed.py:278-299).  -> (path of the best checkpoint, its objective).  Collective calls are made by every rank in the same order:
    the per-step gradient exchange (inside the driver), the BN average, one broadcast of rank 0's decision per validation."""
    import time
    from ._lightning_standin import move_to_device
    log = log or (lambda msg: print(msg, flush=True) if rank == 0 else None)
    tr = task.hparams["training"]
    os.makedirs(log_dir, exist_ok=True)
    task.to(device)
    if use_graph is None:
        use_graph = device.type == "cuda"
    if use_graph:
        from .graph import GraphedStepDriver
        driver = GraphedStepDriver(task, world_size=world, prefetch=os.environ.get("SED_PREFETCH", "teacher"))
    else:
        driver = StepDriver(task, world_size=world, prefetch=os.environ.get("SED_PREFETCH", "teacher"))
    sampler = task.train_sampler
    # a plain DataLoader over the rank's share (pinned host memory: the upload of batch k + 2 runs beside step k)
    loader = torch.utils.data.DataLoader(task.train_data, batch_sampler=sampler, num_workers=task.num_workers, pin_memory=device.type == "cuda")
    n_train = len(loader) if limit_train_batches is None else min(len(loader), int(limit_train_batches))
    if n_train < 1:
        raise SystemExit("no training batch per rank: %d batches of the sampler / %d ranks" % (len(sampler.batch_sampler), world))
    best_metric, best_path = (best if best is not None else (None, None))
    patience, bad_epochs = tr.get("early_stop_patience"), 0
    val_every = int(tr.get("validation_interval", 1) or 1)
    for epoch in range(start_epoch, n_epochs):
        task.train()
        try:
            task.current_epoch = epoch
        except AttributeError:                   # (real Lightning: a read-only property of the attached trainer)
            pass
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        t0 = time.perf_counter()
        it = iter(loader)
        ahead = []                               # host batches k + 1, k + 2 (uploaded early: SEDTask4._stage)

        def pull(k):
            while len(ahead) < 2 and k + 1 + len(ahead) < n_train:
                b = next(it, None)
                if b is None:
                    break
                ahead.append(b)
                task._stage((epoch, k + len(ahead)), b, device)

        cur = move_to_device(next(it), device)
        loss = None
        for k in range(n_train):
            pull(k)
            nxt = None
            if ahead:
                ahead.pop(0)
                nxt = task._take_staged((epoch, k + 1), device)
            loss = driver.run_step(cur, k, next_batch=nxt)
            cur = nxt
            if cur is None:
                break
        del it
        steps = k + 1
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        clips = steps * int(sum(tr["batch_size"])) * world
        log("epoch %d: %d steps x %d ranks, loss %.4f, %.1f clips/s" % (epoch, steps, world, float(loss.detach()), clips / max(dt, 1e-9)))
        if (epoch + 1) % val_every and epoch + 1 < n_epochs:
            continue
        # ---- validation on the rank-averaged BatchNorm statistics; rank 0 scores, everyone learns the verdict ----
        average_bn_buffers(task, world)
        verdict = torch.zeros(3, dtype=torch.float64)                    # (objective, is_best, stop)
        if rank == 0:
            _run_eval(task, task.val_dataloader(), device, task.validation_step, limit_val_batches)
            obj = float(task.validation_epoch_end([]))
            improved = best_metric is None or obj > best_metric
            bad_epochs = 0 if improved else bad_epochs + 1
            verdict[0], verdict[1] = obj, float(improved)
            verdict[2] = float(patience is not None and bad_epochs >= int(patience))
            ckpt = checkpoint_dict(task, epoch + 1)
            ckpt["callbacks"] = {"launcher": {"best_metric": obj if improved else best_metric, "bad_epochs": bad_epochs}}
            torch.save(ckpt, os.path.join(log_dir, "last.ckpt"))
            if improved:
                best_path = os.path.join(log_dir, "epoch=%d-step=%d.ckpt" % (epoch, ckpt["global_step"]))
                if best_metric is not None:
                    for f in os.listdir(log_dir):
                        if f.startswith("epoch=") and f.endswith(".ckpt"):
                            os.remove(os.path.join(log_dir, f))           # save_top_k = 1
                torch.save(ckpt, best_path)
            log("epoch %d: val/obj_metric %.4f%s" % (epoch, obj, " (best)" if improved else ""))
        if world > 1:
            v = verdict.to(device) if dist.get_backend() != "gloo" else verdict
            dist.broadcast(v, src=0)
            verdict = v.cpu()
        if verdict[1] > 0:
            best_metric = float(verdict[0])
        task.reset_pipeline() if hasattr(task, "reset_pipeline") else None
        if verdict[2] > 0:
            log("early stopping after epoch %d" % epoch)
            break
    return best_path, best_metric


def run_test(task, device, limit_test_batches=None):
    """`trainer.test` (train_sed.py:305-306) on this rank."""
    _run_eval(task, task.test_dataloader(), device, task.test_step, limit_test_batches)
    return task.on_test_epoch_end()


def _self_launch(n, argv, module="desed_task_amd.launcher"):
    """`--gpus N` from a plain shell: re-execute this command line under torch.distributed.run, one process per GPU of this node,
    rendezvous on 127.0.0.1 at a free port (the contract bench.py follows)."""
    import socket
    import subprocess
    import sys
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", module] + list(argv)
    sys.stderr.write("desed_task_amd.launcher: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def main(argv=None, cpu_test_device=False, entry_module="desed_task_amd.launcher"):
    """cpu_test_device / entry_module: how tests/launcher_emu.py (the CPU plumbing check: a gloo group on a CPU device, the caller having
    bound its own build of the C-ABI) re-enters this function in every rank; the product command line never sets them."""
    import argparse
    import sys
    import yaml
    ap = argparse.ArgumentParser("python -m desed_task_amd.launcher", description="Data-parallel training of the DESED CRNN mean-teacher baseline "
                                 "on MI355X: train_sed.py's arguments, one process per GPU")
    ap.add_argument("--conf_file", default="./confs/default.yaml")
    ap.add_argument("--log_dir", default="./exp/2023_baseline")
    ap.add_argument("--strong_real", action="store_true")
    ap.add_argument("--resume_from_checkpoint", default=None)
    ap.add_argument("--test_from_checkpoint", default=None)
    ap.add_argument("--eval_from_checkpoint", default=None)
    ap.add_argument("--gpus", default="1", help="number of GPUs of this node (one process each)")
    ap.add_argument("--fast_dev_run", action="store_true")
    args = ap.parse_args(argv)
    n = int(args.gpus)
    if "RANK" not in os.environ and max(n, 1) > 1:
        return _self_launch(n, sys.argv[1:] if argv is None else argv, entry_module)
    if not cpu_test_device and not torch.cuda.is_available():
        raise SystemExit("desed_task_amd.launcher needs the MI355X (no GPU visible); there is no CPU path")
    rank, local, world = init_distributed(backend="gloo" if cpu_test_device else None)
    device = torch.device("cpu") if cpu_test_device else torch.device("cuda", local)
    with open(args.conf_file) as f:
        config = yaml.safe_load(f)
    evaluation = args.eval_from_checkpoint is not None
    test_ckpt = args.eval_from_checkpoint or args.test_from_checkpoint
    log_dir = os.path.join(args.log_dir, "version_0")
    if evaluation:
        config["training"]["batch_size_val"] = 1
    seed = config["training"]["seed"]
    if seed:
        _seed_all(int(seed), rank)
    if test_ckpt is not None:                                    # train_sed.py:367-380: no training, one rank scores
        ckpt = torch.load(test_ckpt, map_location="cpu", weights_only=False)
        hp = dict(ckpt["hyper_parameters"])
        hp["data"] = config["data"]
        task = build_run(hp, log_dir, test_only=True, evaluation=evaluation, fast_dev_run=args.fast_dev_run)
        task.load_state_dict(ckpt["state_dict"])
        task.to(device)
        if rank == 0:
            print("loaded model: %s\nat epoch: %s" % (test_ckpt, ckpt.get("epoch")))
            run_test(task, device, 2 if args.fast_dev_run else None)
        if world > 1:
            dist.barrier()
        return 0
    task = build_run(config, log_dir, rank, world, strong_real=args.strong_real, fast_dev_run=args.fast_dev_run)
    task.to(device)
    start, best = 0, None
    if args.resume_from_checkpoint:
        ckpt = load_checkpoint(task, args.resume_from_checkpoint, resume=True)
        start = int(ckpt.get("epoch", 0))
        cb = (ckpt.get("callbacks") or {}).get("launcher") or {}
        if cb.get("best_metric") is not None:
            best = (float(cb["best_metric"]), None)
    if args.fast_dev_run:
        n_epochs, lim_train, lim_val, lim_test = 3, 2, 2, 2
    else:
        n_epochs, lim_train, lim_val, lim_test = config["training"]["n_epochs"], None, None, None
    best_path, best_metric = fit(task, device, n_epochs, log_dir, world, rank, start, lim_train, lim_val, best=best)
    if rank == 0:
        print("best model: %s (val/obj_metric %s)" % (best_path, best_metric))
        if best_path is not None:
            task.load_state_dict(torch.load(best_path, map_location="cpu", weights_only=False)["state_dict"])
        run_test(task, device, lim_test)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
