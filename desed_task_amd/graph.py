"""Whole-step HIP graphs: capture one mean-teacher training step once, replay it every step.

A step is ~330 kernel launches issued from one Python thread; on a busy host the enqueue time (3.5-8 ms) rivals the
GPU time (~6 ms).  Capturing the step (both forwards, the losses, the EMA, backward, Adam) into ONE hipGraph makes the
step time independent of the host.  What a capture freezes is every by-value launch argument, so everything that
changes from step to step lives in a small DEVICE buffer (`DynArgs`) that the kernels read through the nullable
`*_dev` pointers of include/sed_hip.h:

    dropout seeds (one per dropout call site), mixup coefficients + permutations, consistency-loss weight,
    EMA factor, Adam step size / bias correction.

The host half of a step -- RNG draws in the reference's order, the lr/rampup schedule, step counters -- is recorded
during the capture pass as a list of closures (`DynArgs.host`) and re-run before every replay; it fills a pinned host
mirror that is uploaded with one async copy.  The capture pass itself IS a real step: its host half runs inline,
the device half runs at the first replay.  `GraphedStepDriver` keeps Lightning's step order (launcher.StepDriver).

On a CPU "device" (the fiber emulator used by the tests) DynArgs aliases the host buffer as the device buffer and no
graph is involved: the same kernels, reading their step-varying arguments from memory, can be checked against the
plain eager path.
"""
import torch

_active = None


def active():
    """The DynArgs the current step runs under (None = plain eager launches with by-value arguments)."""
    return _active


class DynSeed(int):
    """A dropout seed that lives in device memory: int value 0 (the by-value part), `.dev` = address of the seed word."""
    dev = None


class DynFloat(float):
    """A scalar that lives in device memory: float value = the capture-time value, `.dev` = address, `.tensor` = 0-d view."""
    dev = None
    tensor = None


def seed_dev(seed):
    return getattr(seed, "dev", None)


class DynArgs:
    F_LOSS_W, F_EMA_ALPHA, F_EMA_OMA, F_ADAM_STEP, F_ADAM_IBC2, F_MIX_C0 = 0, 1, 2, 3, 4, 5   # float slots; {c, 1-c} per group
    RING = 4
    SEED0, N_SEEDS = 24, 40
    PERM0, PERM_LEN, N_PERMS = 64, 64, 8            # 8 mixup sites: the 2024 step mixes 3 data sets x (features, embeddings)

    def __init__(self, device):
        n = self.PERM0 + self.PERM_LEN * self.N_PERMS
        device = torch.device(device)
        self.device = device
        self._ring, self._ring_ev, self._ring_i = None, None, 0
        if device.type == "cuda":
            self.hbuf = torch.zeros(n, dtype=torch.int32).pin_memory()
            self.dev = torch.zeros(n, dtype=torch.int32, device=device)
            # The upload is asynchronous and a replay returns at once, so the host may be several steps ahead of the GPU: the copy of
            # step k must not read a buffer the host half of step k + 1 is already rewriting.  Each upload therefore goes out of its
            # own pinned slot of a small ring; a slot is reused only after its previous copy has executed (event).
            self._ring = [torch.zeros(n, dtype=torch.int32).pin_memory() for _ in range(self.RING)]
            self._ring_ev = [None] * self.RING
        else:
            self.hbuf = torch.zeros(n, dtype=torch.int32)
            self.dev = self.hbuf
        self.hf = self.hbuf.view(torch.float32)
        self.df = self.dev.view(torch.float32)
        self.ops = []               # host half of the step, in program order
        self.recording = False
        self.state = {}
        self._sites = {"seed": 0, "mix": 0}

    # ---- addresses ---------------------------------------------------------------------------------
    def ptr(self, slot):
        return self.dev.data_ptr() + 4 * slot

    def begin_step(self):
        self._sites = {"seed": 0, "mix": 0}

    # ---- host half -----------------------------------------------------------------------------------
    def host(self, fn):
        """Run a piece of the step's host logic now; while recording, keep it for the replays."""
        fn()
        if self.recording:
            self.ops.append(fn)

    def run_host_ops(self):
        for fn in self.ops:
            fn()

    def upload(self):
        if self.dev is self.hbuf:
            return
        i = self._ring_i
        self._ring_i = (i + 1) % self.RING
        if self._ring_ev[i] is not None:
            self._ring_ev[i].synchronize()           # the copy that last read this slot (RING steps ago) has executed
        self._ring[i].copy_(self.hbuf)               # host -> host, synchronous, 2.3 KB
        # (round 4: a kernel on the step's own stream reading the pinned slot instead of this runtime copy -- which goes out as a blit
        #  on another hardware queue -- was built and measured: 3.153 / 3.146 vs 3.161 / 3.144 ms per step, same box: neutral, removed)
        self.dev.copy_(self._ring[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_ev[i] = ev

    # ---- call sites ------------------------------------------------------------------------------------
    def new_seed(self, draw):
        """One dropout call site: the seed word is drawn on the host by `draw()` every step, read by the kernel from HBM."""
        i = self._sites["seed"]
        if i >= self.N_SEEDS:
            raise RuntimeError("DynArgs: more dropout call sites than seed slots")
        self._sites["seed"] = i + 1
        slot = self.SEED0 + i
        host = self.hbuf

        def fill():
            host[slot] = int(draw())                    # 31-bit draw; the kernels add the word to the by-value seed 0

        self.host(fill)
        s = DynSeed(0)
        s.dev = self.ptr(slot)
        s.slot = slot
        return s

    def seed_value(self, seed):
        """The effective 32-bit seed of a DynSeed for the CURRENT host buffer contents (tests)."""
        return int(self.hbuf[seed.slot])

    def scalar(self, slot, compute, complement=False):
        """A float argument recomputed by `compute()` on the host every step (complement: slot + 1 <- 1 - value, in the
        same double -> float32 rounding the by-value path uses)."""
        hf = self.hf

        def fill():
            v = float(compute())
            hf[slot] = v
            if complement:
                hf[slot + 1] = 1.0 - v

        self.host(fill)
        v = DynFloat(float(hf[slot]))
        v.dev = self.ptr(slot)
        v.tensor = self.df[slot]
        return v

    def mix_site(self, n, draw):
        """One mixup group of n clips: `draw()` -> (c, perm) or None (= no mixup this step: sentinel c = 2, identity)."""
        g = self._sites["mix"]
        if g >= self.N_PERMS or n > self.PERM_LEN:
            raise RuntimeError("DynArgs: mixup group does not fit the permutation slots")
        self._sites["mix"] = g + 1
        cslot, pslot = self.F_MIX_C0 + 2 * g, self.PERM0 + g * self.PERM_LEN
        hf, host = self.hf, self.hbuf
        ident = torch.arange(n, dtype=torch.int32)

        def fill():
            r = draw()
            if r is None:
                hf[cslot] = 2.0                 # sentinel outside [0, 1]: the mixup launches of this replay return at once
                hf[cslot + 1] = 0.0
                host[pslot:pslot + n] = ident
            else:
                hf[cslot] = float(r[0])
                hf[cslot + 1] = 1.0 - float(r[0])
                host[pslot:pslot + n] = r[1].to(torch.int32)

        self.host(fill)
        return self.ptr(cslot), self.ptr(pslot)


class dyn_step:
    """Context manager: run one step's Python under `dyn` (kernels take their step-varying arguments from HBM)."""

    def __init__(self, dyn, record=False):
        self.dyn, self.record = dyn, record

    def __enter__(self):
        global _active
        self.prev = _active
        _active = self.dyn
        self.dyn.begin_step()
        if self.record:
            self.dyn.ops = []
        self.dyn.recording = self.record
        return self.dyn

    def __exit__(self, *exc):
        global _active
        _active = self.prev
        self.dyn.recording = False
        return False


def quiesce_collectives(dev):
    """Called right before a stream capture: drain the device AND the collective backend's watchdog.

    ProcessGroupNCCL keeps every collective it issued in a work list that its watchdog thread polls with hipEventQuery every 100 ms
    until the work has completed.  On this ROCm stack such a query from ANOTHER thread while this thread captures the step (three
    side streams forked and joined inside the capture) intermittently fails, `capture_error_mode="thread_local"` notwithstanding:
    the watchdog's exception aborts the process, or the capture is invalidated (the next launch returns
    hipErrorStreamCaptureInvalidated) -- 1 - 7 % of the captures; single-stream captures survived the same polls.  That, not a numeric race, was round 3's
    "one run in 43" of the one-rank RCCL rehearsal: tools/rehearsal_loop.py reproduced it as 6 dead processes in 87 repetitions,
    every survivor bit-identical.  With the device idle, `_wait_for_pending_works()` returns as soon as the watchdog has retired
    the last work item; after that it has no event left to query until the next collective -- and none is ever issued during a
    capture (launcher.StepDriver keeps all collectives outside the graphs)."""
    torch.cuda.synchronize(dev)
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    groups = list(getattr(dist.distributed_c10d._world, "pg_map", {}).keys()) or [dist.distributed_c10d._get_default_group()]
    waited = True
    for pg in groups:
        try:
            nccl = dist.get_backend(pg) == "nccl"
        except Exception:  # noqa: BLE001 -- a group this rank is not part of
            continue
        if nccl:
            wait = getattr(pg, "_wait_for_pending_works", None)
            if wait is None:
                waited = False
            else:
                wait()
    if not waited:
        # (a torch build without ProcessGroup._wait_for_pending_works: the device is idle, so every work item is complete and the
        #  watchdog retires it at its next 100-ms poll -- give it three)
        import time
        import warnings
        warnings.warn("graph.quiesce_collectives: this torch build has no ProcessGroup._wait_for_pending_works; falling back to a 0.3 s "
                      "pause before the stream capture.  If the RCCL watchdog still invalidates the capture it fails loudly (hipErrorStreamCapture"
                      "Invalidated / an aborted process): it is NOT retried -- the capture pass runs the step's host half (RNG draws, "
                      "scheduler, Adam's step count), which a second attempt would run twice")
        time.sleep(0.3)


class GraphedStepDriver:
    """launcher.StepDriver with the step captured in a hipGraph (see the module docstring).

    The first `warmup` steps run eagerly (they are ordinary training steps and let the allocator, the lazily built
    buffers and hipFuncSetAttribute calls settle); the next step is captured; every later step re-runs the recorded
    host half, uploads DynArgs, copies the batch into the static input buffers and replays.

    Every step -- eager warm-up, capture, replay -- runs on the driver's OWN stream: autograd's AccumulateGrad nodes are
    bound to the stream of the first backward and live as long as the parameters, and a node bound to the legacy default
    stream cannot take part in a capture (hipStreamEndCapture faults).  The caller's current stream is joined on entry
    and exit, so callers need no extra synchronisation.

    world_size > 1: no collective is ever captured.  The step is TWO graphs -- [forwards, losses, EMA, backward of the heads and
    the BiGRU] and [backward of the CNN] -- with the asynchronous all-reduce of gradient bucket A issued between them (it runs on
    RCCL's stream under the second graph), then bucket B's all-reduce and the Adam launch, eager.  With SED_DDP_OVERLAP=0: one
    graph up to the end of backward, one blocking all-reduce over the arena, Adam."""

    def __init__(self, task, world_size=1, warmup=3, ema_side_stream=True, prefetch=None):
        from .launcher import StepDriver
        self.eager = StepDriver(task, world_size, ema_side_stream=ema_side_stream, prefetch=prefetch)
        self.static_next = None     # pipelined front-end: static buffer of the NEXT batch's waveforms (the graph's mel branch reads it)
        self.static_next_labels = None      # ... and, prefetch "teacher", of its labels (read by the graph's side branch)
        self.static_next_extras = {}        # ... and of whatever else of the next batch the front half reads (2024: the embeddings)
        self.task = task
        self.world = world_size
        self.warmup = warmup
        self.n = 0
        self.graph = None
        self.graph_cnn = None       # world_size > 1, overlapped exchange: the CNN half of backward is its own graph
        self.dyn = None
        self.static = None
        self.loss = None
        self.stream = None
        # SED_DDP_GRAPH_EXCHANGE=1 (round 5): capture the gradient exchange TOO -- the RCCL all-reduce(s) and the
        # Adam launch become nodes of the one graph, like the single-GPU step: no second graph, no host hand-over between the replay
        # and the collective (measured 18 us device / 61 us host with a no-op collective).  A captured collective never reaches the
        # process group's watchdog, so the capture-invalidation race of round 3 cannot recur for it.  RCCL only (the gloo test backend
        # synchronises the device around its collectives); rehearsed on one rank -- never yet on a real multi-GPU node, hence opt-in.
        import os
        self.capture_exchange = os.environ.get("SED_DDP_GRAPH_EXCHANGE") == "1" and self.eager.exchange
        self.eager_fallbacks = 0    # pipelined graph: steps run eagerly because no next batch was announced (epoch ends)
        self.reprimes = 0           # ... and front halves run inline because the previous step had prefetched nothing

    def _primed(self):
        """Pipelined front end: has the previous step left this step's front half in the task's hand-over buffers?  (A replay runs no
        Python: after the capture the task's flags stay set until an eager step consumes them or reset_pipeline() clears them.)"""
        t = self.task
        return bool(getattr(t, "_feat_ready", False) or (getattr(t, "_pro", None) or {}).get("ready"))

    def _reprime(self, batch, teacher_level):
        """Run the front half of `batch` now (eagerly, on the prefetch stream, joined): the captured step that follows consumes it
        exactly as if the previous step had prefetched it.  The host draws it makes (mixup coin / c / permutations, the teacher
        CNN's seeds) are the ones the unpipelined order makes at the head of this step."""
        task = self.task
        if teacher_level and (len(batch) < 2 or batch[1] is None):
            raise ValueError('prefetch "teacher": the batch must carry its labels')
        extras = task.next_batch_extras(batch) if (teacher_level and hasattr(task, "next_batch_extras")) else {}
        pairs = [(batch[0], self.static_next), (batch[1] if teacher_level else None, self.static_next_labels)]
        pairs += [(extras[k], self.static_next_extras[k]) for k in self.static_next_extras]
        for src, dst in pairs:
            if src is not None and src.data_ptr() != dst.data_ptr():
                if src.shape != dst.shape:
                    raise ValueError("batch tensor shapes changed after the step was captured")
                dst.copy_(src, non_blocking=True)
        task.set_next_batch(self.static_next, self.static_next_labels if teacher_level else None, dict(self.static_next_extras))
        task.launch_prefetch(task.prefetch_point)
        task.join_prefetch()
        self.eager._announced = None            # (whatever was announced before the pipeline was reset is void)
        self.reprimes += 1

    def _device(self):
        return next(self.task.sed_student.parameters()).device

    def _freeze_handover_buffers(self):
        """From the capture on, the graph reads the task's hand-over buffers (`_pro`, `_feat_buf`) at FIXED addresses: a silent
        reallocation -- another shape, stride or dtype while nothing is ready -- would leave the replays on the old storage (ADVICE
        r04).  Frozen buffers raise instead."""
        pro = getattr(self.task, "_pro", None)
        if pro is not None:
            pro["frozen"] = True

    def input_buffers(self):
        """The static device tensors the captured graph reads (one per tensor of the batch tuple, None elsewhere), available
        after the capture step.  A data pipeline that writes its batches straight into them (e.g. as the target of its
        host-to-device copies) and passes them to run_step() saves the per-step device-to-device staging copy; tensors the step
        modifies in place (the labels under mixup) must of course be rewritten every step."""
        return self.static

    def next_label_buffer(self):
        """prefetch "teacher": the static buffer of the NEXT batch's labels (only read by the graph's side branch, which copies them
        into the step's hand-over buffer and mixes them there)."""
        return self.static_next_labels

    def next_extra_buffers(self):
        """prefetch "teacher": {name: static buffer} of the other next-batch tensors the front half reads (2024: "embeddings")."""
        return self.static_next_extras

    def next_audio_buffer(self):
        """Pipelined front-end: the static buffer the graph's prefetch branch reads the NEXT batch's waveforms from (None when off
        or before the capture).  A loader that writes there and passes it as next_batch[0] saves the staging copy."""
        return self.static_next

    def _step_body(self, batch):
        """One step in Lightning's order.  world_size 1: the whole step.  world_size > 1: up to and including loss.backward() --
        which, with the overlapped gradient exchange, ends at the cut behind the student's CNN (launcher.StepDriver)."""
        d = self.eager
        task = self.task
        d.arm_overlap()
        from . import ops as _ops
        _ops.probe("step_start")
        loss = d.training_step_and_ema(batch, 0)         # (the EMA and the parked loss sums on the side stream)
        _ops.probe("loss_done")
        d.opt.zero_grad(set_to_none=True)
        from . import launcher as _launcher
        late = _launcher.PREFETCH_ENQUEUE_LATE
        fork = None
        if late:
            fork = torch.cuda.Event()
            fork.record()
        else:
            task.launch_prefetch("backward", after=(d.side,))
        if self.capture_exchange:
            d.backward(loss)                             # ... + the CNN half + the all-reduce(s), all of it inside the capture
        else:
            d.backward_joined(loss)                      # BiGRU weight-gradient GEMMs on the side stream, joined here
        _ops.probe("backward_done")
        if late:
            task.launch_prefetch("backward", after=(d.side,), fork_event=fork)
        task.join_prefetch()
        if d.side is not None:
            torch.cuda.current_stream().wait_stream(d.side)
        if not d.exchange or self.capture_exchange:
            d.opt.step()
            task.lr_scheduler_step(d.sched, 0, None)
        _ops.probe("step_end")
        return loss

    def _finish_multi(self):
        """Tail of a data-parallel step around the second graph: [bucket A all-reduce, async] -> replay of the CNN backward ->
        bucket B all-reduce -> Adam + scheduler (eager, by-value arguments).  No collective is ever captured."""
        d = self.eager
        d.bucket_log = []               # (arm_overlap() runs at the capture only: the log is the LAST step's collectives)
        d._mark("backward_done")        # (two graphs: of the heads + BiGRU -- bucket A is complete)
        if self.graph_cnn is not None:
            d.launch_bucket_a()
            self.graph_cnn.replay()
            d.finish_buckets()
        else:
            d.allreduce_grads()
        d._mark("exchange_done")
        d.opt.step()
        d._mark("adam_done")
        self.task.lr_scheduler_step(d.sched, 0, None)

    def run_step(self, batch, batch_idx=0, next_batch=None):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("GraphedStepDriver needs a GPU (use StepDriver, or graph.dyn_step for CPU checks)")
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        caller = torch.cuda.current_stream(dev)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            loss = self._run_step(batch, batch_idx, dev, next_batch)
        caller.wait_stream(self.stream)
        return loss

    def _run_step(self, batch, batch_idx, dev, next_batch=None):
        self.n += 1
        if self.n <= self.warmup:
            return self.eager.run_step(batch, batch_idx, next_batch)
        task = self.task
        pipelined = getattr(task, "prefetch_point", None) is not None
        teacher_level = pipelined and getattr(task, "prefetch_level", "features") == "teacher"
        if pipelined and self.graph is not None:
            # The captured step has both halves of the pipeline baked in: it CONSUMES the front half prefetched by the previous step
            # and PRODUCES the next one from the static next-batch buffers.  Two situations break that chain:
            #  * nothing is announced (last batch of an epoch): a replay would run its side branch on the stale contents of the
            #    static buffers -- consuming host draws and moving the teacher's BatchNorm statistics for a batch nobody trains on.
            #    That step runs EAGERLY instead (same kernels, by-value arguments, no side branch), like launcher.StepDriver does;
            #  * nothing was prefetched for this batch (the step after such an eager step, or after reset_pipeline(): weights were
            #    loaded in between): the front half of THIS batch is run inline first (`_reprime`), then the replay proceeds.
            if next_batch is None:
                self.eager_fallbacks += 1
                return self.eager.run_step(batch, batch_idx, None, staged=self.static_next)
            if not self._primed():
                self._reprime(batch, teacher_level)
        if pipelined and self.graph is None and (next_batch is None or not self._primed()):
            # the capture needs both halves of the pipeline in hand: a successor to announce and a front half an eager step left for
            # this batch.  An epoch end inside the warm-up (or a loop that cannot announce) postpones it: one more eager step
            return self.eager.run_step(batch, batch_idx, next_batch)
        if pipelined:
            self.eager.announce(batch, next_batch, staged=self.static_next)
            nxt, nxt_lab, nxt_ext = task._next_audio, task._next_labels, (task._next_extras or {})
            if self.graph is None:
                ready = task._feat_ready or (task._pro is not None and task._pro["ready"])
                if nxt is None or not ready or (teacher_level and nxt_lab is None):
                    raise RuntimeError("pipelined front-end: the capture step needs a front half prefetched by an eager step before it "
                                       "(warmup >= 1) and a next_batch")
                self.static_next = torch.empty_like(nxt)
                if teacher_level:
                    self.static_next_labels = torch.empty_like(nxt_lab)
                    self.static_next_extras = {k: torch.empty_like(v) for k, v in nxt_ext.items()}
            pairs = [(nxt, self.static_next), (nxt_lab if teacher_level else None, self.static_next_labels)]
            if teacher_level:
                if set(nxt_ext) != set(self.static_next_extras):
                    raise ValueError("the announced batch's extra tensors changed after the step was captured")
                pairs += [(nxt_ext[k], self.static_next_extras[k]) for k in self.static_next_extras]
            for src, dst in pairs:
                if src is not None and src.data_ptr() != dst.data_ptr():
                    if src.shape != dst.shape:
                        raise ValueError("next_batch tensor shapes changed after the step was captured")
                    dst.copy_(src, non_blocking=True)
            if self.graph is None:
                task.set_next_batch(self.static_next, self.static_next_labels, dict(self.static_next_extras))
            else:
                task.set_next_batch(None, None)
        if self.graph is None:
            if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            self.dyn = DynArgs(dev)
            # every device tensor of the batch (audio, labels, and e.g. the embeddings of the pretrained recipe) gets a static
            # input buffer: the graph reads those addresses, each replay copies the new batch into them
            self.static = tuple(torch.empty_like(t) if torch.is_tensor(t) and t.is_cuda else None for t in batch)
            for st, t in zip(self.static, batch):
                if st is not None:
                    st.copy_(t)
            B = batch[0].shape[0] if torch.is_tensor(batch[0]) else 0
            if B:
                from . import ops as _ops
                _ops.loss_work(dev, B)              # (scratch the loss kernel caches per device: allocated OUTSIDE the capture pool)
            quiesce_collectives(dev)
            self.graph = torch.cuda.CUDAGraph()
            self._freeze_handover_buffers()
            with dyn_step(self.dyn, record=True):
                # thread_local: calls other threads make (allocator, a collective backend's housekeeping) must not fail this capture;
                # the one that does fail on this stack -- the RCCL watchdog's event query -- has been drained by quiesce_collectives()
                with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                    self.loss = self._step_body(tuple(st if st is not None else t for st, t in zip(self.static, batch)))
                student = self.task.sed_student
                if self.eager.exchange and not self.capture_exchange and getattr(student, "_cnn_boundary", None) is not None:
                    # the cut left the CNN half of backward undone: it becomes a second graph (same memory pool), replayed
                    # after bucket A has been handed to RCCL
                    self.graph_cnn = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_cnn, pool=self.graph.pool(), stream=self.stream, capture_error_mode="thread_local"):
                        student.backward_cnn()
        else:
            if len(batch) != len(self.static):
                raise ValueError("batch arity changed after the step was captured")
            skip = getattr(task, "prefetched_batch_fields", (1,)) if teacher_level else ()
            for st, t in zip(self.static, batch):
                if st is not None:
                    if not torch.is_tensor(t) or t.shape != st.shape:
                        raise ValueError("batch tensor shapes changed after the step was captured")
                    if pipelined and st is self.static[0]:
                        continue                            # the graph reads this batch's FEATURES (prefetched), not its waveforms
                    if teacher_level and any(st is self.static[i] for i in skip if i < len(self.static)):
                        continue                            # ... and its labels (+ the 2024 step's embeddings) as the previous
                                                            # replay's side branch left them in the hand-over buffers
                    if t.data_ptr() != st.data_ptr():       # a loader may fill the static buffers directly (input_buffers())
                        st.copy_(t, non_blocking=True)
            self.dyn.run_host_ops()
        self.dyn.upload()
        probe = self.eager.probe
        if probe is not None:
            probe.begin()
        self.graph.replay()
        if self.eager.exchange and not self.capture_exchange:
            self._finish_multi()
        if probe is not None:
            probe.end()
        return self.loss
