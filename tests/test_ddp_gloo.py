"""Data-parallel path on CPU: world_size 2, gloo, kernels through the fiber emulator.
Checks the StepDriver contract: each rank runs the single-GPU step on its own clips, the flat gradient arena is
summed with ONE all-reduce, the 1/world factor is folded into Adam, and the student stays bit-identical
across ranks while BN statistics stay rank-local."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tests.emu_support import bind_emulator
    bind_emulator()
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd.launcher import StepDriver, init_distributed
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    bs, n_samp = (1, 1, 1), 8192 + 1024
    sd = O.make_state_dict(seed=7)
    task = P.build_task("cpu", bs, sd, dropout=0.0, specaug=False, rampup=100)
    driver = StepDriver(task, world_size=world)
    audio = O.synth_audio(3, n_samp, seed=100 + rank)           # different clips per rank
    n_out = (1 + n_samp // 256) // 4
    labels = O.synth_labels(bs, 10, n_out, seed=5 + rank)
    import random
    random.seed(4); np.random.seed(7); torch.manual_seed(7)     # same mixup gate everywhere, draws rank-local anyway
    # local gradient (before the exchange), for the averaging check
    loss = task.training_step((audio.clone(), labels.clone(), None, None), 0)
    task.opt.zero_grad(set_to_none=True)
    loss.backward()
    local = task.sed_student.arena.gather_grads().clone()
    task.opt.zero_grad(set_to_none=True)
    # the real step
    random.seed(4); np.random.seed(7); torch.manual_seed(7)
    task2 = P.build_task("cpu", bs, sd, dropout=0.0, specaug=False, rampup=100)
    driver = StepDriver(task2, world_size=world)
    assert driver.overlap                                       # default: bucketed exchange, bucket A under the CNN backward
    driver.run_step((audio.clone(), labels.clone(), None, None), 0)
    arena = task2.sed_student.arena
    summed = arena.flat_grad.clone()
    # bucket boundaries: A = BiGRU + heads, launched first; B = the CNN
    names = [n for n, _ in task2.sed_student.named_parameters()]
    split = arena.offsets[[i for i, n in enumerate(names) if not n.startswith("cnn.")][0]]
    assert driver.bucket_log == [("A", split, arena.numel - split), ("B", 0, split)], driver.bucket_log
    assert abs(4 * (arena.numel - split) / 1e6 - 2.00) < 0.01 and abs(4 * split / 1e6 - 2.45) < 0.01
    assert task2.sed_student._cnn_boundary is None
    # the single blocking all-reduce over the whole arena gives the same sums, bit for bit
    random.seed(4); np.random.seed(7); torch.manual_seed(7)
    task3 = P.build_task("cpu", bs, sd, dropout=0.0, specaug=False, rampup=100)
    driver3 = StepDriver(task3, world_size=world, overlap_allreduce=False)
    driver3.run_step((audio.clone(), labels.clone(), None, None), 0)
    assert driver3.bucket_log == [("AB", 0, arena.numel)]
    assert torch.equal(task3.sed_student.arena.flat_grad, summed) and torch.equal(task3.sed_student.arena.flat, arena.flat)
    # broadcast_state: a rank that starts from other weights / BN buffers is pulled onto rank 0's
    sd_other = O.make_state_dict(seed=7 + rank)
    task4 = P.build_task("cpu", bs, sd_other, dropout=0.0, specaug=False, rampup=100)
    StepDriver(task4, world_size=world)
    ref4 = P.build_task("cpu", bs, O.make_state_dict(seed=7), dropout=0.0, specaug=False, rampup=100)
    assert torch.equal(task4.sed_student.arena.flat, ref4.sed_student.arena.flat)
    assert torch.equal(task4.sed_teacher.cnn.cnn.batchnorm2.running_var, ref4.sed_teacher.cnn.cnn.batchnorm2.running_var)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    flats = [torch.zeros_like(arena.flat) for _ in range(world)]
    dist.all_gather(flats, arena.flat.clone())
    rm = task2.sed_student.cnn.cnn.batchnorm0.running_mean.clone()
    rms = [torch.zeros_like(rm) for _ in range(world)]
    dist.all_gather(rms, rm)
    # SURVEY 8e policy: BN buffers are averaged before validation / checkpointing, the checkpoint is written by rank 0
    from desed_task_amd.launcher import average_bn_buffers, bn_buffers, save_checkpoint
    rv_t = task2.sed_teacher.cnn.cnn.batchnorm3.running_var.clone()
    rvs_t = [torch.zeros_like(rv_t) for _ in range(world)]
    dist.all_gather(rvs_t, rv_t)
    assert len(bn_buffers(task2)) == 2 * 2 * 7 and sum(b.numel() for b in bn_buffers(task2)) == 2 * 1248
    ckpt = os.path.join(out_dir, "ckpt.pt")
    save_checkpoint(task2, ckpt, epoch=3)
    assert os.path.exists(ckpt)                                   # every rank returns after rank 0 wrote it
    rm_avg = task2.sed_student.cnn.cnn.batchnorm0.running_mean.clone()
    rv_avg_t = task2.sed_teacher.cnn.cnn.batchnorm3.running_var.clone()
    # resume: a fresh task that loads the checkpoint continues exactly like the one that wrote it (weights, BN buffers, Adam
    # moments + step, scheduler step_num -> lr, consistency weight, EMA factor)
    from desed_task_amd.launcher import load_checkpoint
    task5 = P.build_task("cpu", bs, O.make_state_dict(seed=11), dropout=0.0, specaug=False, rampup=100)
    ck = load_checkpoint(task5, ckpt)
    assert ck["epoch"] == 3 and task5.scheduler["scheduler"].step_num == task2.scheduler["scheduler"].step_num == 2
    driver5 = StepDriver(task5, world_size=world, broadcast_init=False)
    from tests.emu_support import emu_threads
    prev_threads = emu_threads(1)        # bit-exact comparison of two runs: in-order workgroups (fp32 atomics in a fixed order)
    for t_, d_ in ((task2, driver), (task5, driver5)):
        random.seed(4); np.random.seed(8); torch.manual_seed(8)
        d_.run_step((audio.clone(), labels.clone(), None, None), 1)
    emu_threads(prev_threads if prev_threads > 0 else min(8, os.cpu_count() or 1))
    assert torch.equal(task5.sed_student.arena.flat, task2.sed_student.arena.flat)
    assert torch.equal(task5.sed_teacher.arena.flat, task2.sed_teacher.arena.flat)
    assert task5.opt.param_groups[0]["lr"] == task2.opt.param_groups[0]["lr"]
    avgs = [torch.zeros_like(rm_avg) for _ in range(world)]
    dist.all_gather(avgs, rm_avg)
    if rank == 0:
        torch.save(dict(summed=summed, gathered=gathered, flats=flats, rms=rms, avgs=avgs, rvs_t=rvs_t, rv_avg_t=rv_avg_t),
                   os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_gradient_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    g0, g1 = d["gathered"]
    assert (g0 - g1).abs().max() > 1e-6                                    # ranks really saw different data
    ref_sum = g0 + g1
    assert (d["summed"] - ref_sum).abs().max().item() <= 1e-6 * ref_sum.abs().max().item() + 1e-9
    f0, f1 = d["flats"]
    assert torch.equal(f0, f1)                                             # student bit-identical across ranks
    assert not torch.equal(d["rms"][0], d["rms"][1])                       # BN running stats stay rank-local ...
    # ... until average_bn_buffers / save_checkpoint: every rank then holds the mean over ranks, student and teacher
    assert torch.equal(d["avgs"][0], d["avgs"][1])
    assert (d["avgs"][0] - (d["rms"][0] + d["rms"][1]) / 2).abs().max().item() < 1e-7
    assert (d["rv_avg_t"] - (d["rvs_t"][0] + d["rvs_t"][1]) / 2).abs().max().item() < 1e-6
    # Lightning-shaped checkpoint: what train_sed.py reads back (:302, :367-374) + the on_save_checkpoint extras (sed_trainer.py:603)
    ck = torch.load(os.path.join(str(tmp_path), "ckpt.pt"), weights_only=False)
    assert {"state_dict", "hyper_parameters", "epoch", "optimizer_states", "lr_schedulers", "global_step", "sed_student",
            "sed_teacher"} <= set(ck)
    assert torch.equal(ck["sed_student"]["cnn.cnn.batchnorm0.running_mean"], d["avgs"][0])
    assert torch.equal(ck["state_dict"]["sed_student.cnn.cnn.batchnorm0.running_mean"], d["avgs"][0])
    assert "sed_teacher.dense.weight" in ck["state_dict"] and ck["hyper_parameters"]["training"]["batch_size"] == [1, 1, 1]
    # torchaudio's persistent buffers of the reference's mel_spec travel too (strict load on the reference side)
    assert tuple(ck["state_dict"]["mel_spec.mel_scale.fb"].shape) == (1025, 128) and "mel_spec.spectrogram.window" in ck["state_dict"]
    ost = ck["optimizer_states"][0]
    assert len(ost["state"]) == 62 and set(ost["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(ost["state"][0]["step"]) == 1.0
    assert ck["lr_schedulers"][0]["step_num"] == 2 and ck["global_step"] == 1


def _multi_step_worker(rank, world, port, out_dir):
    """Three consecutive data-parallel steps (dropout + SpecAugment + mixup on, different clips per rank): at EVERY step the reduced
    gradient arena must be the sum of the ranks' local arenas as they were handed to the collective, and the students must stay
    bit-identical across ranks -- by induction the N-rank run is the single-GPU reference step per rank (rank-local BatchNorm
    statistics, mixup and loss means) with the gradient mean in Adam."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tests.emu_support import bind_emulator
    bind_emulator()
    import random
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver, init_distributed
    init_distributed(backend="gloo")
    bs, n_samp, steps = (1, 1, 1), 4096 + 1024, 3
    sd = O.make_state_dict(seed=7)
    task = P.build_task("cpu", bs, sd, dropout=0.5, specaug=True, rampup=5)
    driver = StepDriver(task, world_size=world, overlap_allreduce=False)       # one all-reduce over the whole arena: one pre / post pair
    n_out = (1 + n_samp // 256) // 4
    random.seed(4); np.random.seed(7 + rank); torch.manual_seed(7 + rank)
    _ops.reseed_dropout()
    real_all_reduce = dist.all_reduce
    record = []

    def spy(tensor, *a, **k):
        pre = tensor.detach().clone()
        out = real_all_reduce(tensor, *a, **k)
        record.append((pre, tensor.detach().clone()))
        return out

    dist.all_reduce = spy
    ok = True
    try:
        for step in range(steps):
            audio = O.synth_audio(3, n_samp, seed=300 + 10 * step + rank)
            labels = O.synth_labels(bs, 10, n_out, seed=40 + 10 * step + rank)
            del record[:]
            driver.run_step((audio, labels, None, None), step)
            assert len(record) == 1 and record[0][0].numel() == task.sed_student.arena.numel
            pre, post = record[0]
            dist.all_reduce = real_all_reduce
            pres = [torch.zeros_like(pre) for _ in range(world)]
            dist.all_gather(pres, pre)
            flats = [torch.zeros_like(pre) for _ in range(world)]
            dist.all_gather(flats, task.sed_student.arena.flat.detach().clone())
            dist.all_reduce = spy
            want = pres[0] + pres[1]
            ok &= bool((pres[0] - pres[1]).abs().max() > 1e-7)                                       # ranks saw different data / masks
            ok &= bool((post - want).abs().max() <= 1e-6 * want.abs().max() + 1e-12)                 # sum of the local gradients
            ok &= bool(torch.equal(flats[0], flats[1]))                                              # same Adam on every rank
    finally:
        dist.all_reduce = real_all_reduce
    if rank == 0:
        torch.save(dict(ok=ok, steps=steps), os.path.join(out_dir, "multi.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_multi_step_semantics(tmp_path):
    mp.spawn(_multi_step_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    d = torch.load(os.path.join(str(tmp_path), "multi.pt"))
    assert d["ok"] and d["steps"] == 3


def _side_worker(rank, world, port, out_dir):
    """Two ranks, the weight-gradient side sections ON under the exchange (round 5: the default over RCCL when the gradients leave as one
    all-reduce; forced here with SED_GRU_DW_SIDE=1 and run through the emulator's parking logic, ops.SIDE_ON_CPU): every parked launch
    -- BiGRU sections, head sums, CNN weight gradients -- must have gone out before the arena is handed to the collective, i.e. two
    steps end in the same bits as with everything on the chain."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from tests.emu_support import bind_emulator
    bind_emulator()
    import random
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver, init_distributed
    init_distributed(backend="gloo")
    bs, n_samp = (1, 1, 1), 4096 + 1024
    sd = O.make_state_dict(seed=7)
    n_out = (1 + n_samp // 256) // 4
    out = {}
    for mode in ("chain", "side"):
        if mode == "side":
            os.environ["SED_GRU_DW_SIDE"] = "1"
            _ops.SIDE_ON_CPU = True
        else:
            os.environ["SED_GRU_DW_SIDE"] = "0"
            _ops.SIDE_ON_CPU = False
        task = P.build_task("cpu", bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = StepDriver(task, world_size=world, overlap_allreduce=False)
        assert driver.exchange and not driver.overlap and driver.gru_dw_side == (mode == "side")
        random.seed(4); np.random.seed(7 + rank); torch.manual_seed(7 + rank)
        _ops.reseed_dropout()
        parked = []
        if mode == "side":
            real = _ops.defer_off_chain

            def spy(device, launch, keep):
                parked.append(len(_ops._deferred))
                return real(device, launch, keep)
            _ops.defer_off_chain = spy
        try:
            for step in range(2):
                audio = O.synth_audio(3, n_samp, seed=300 + 10 * step + rank)
                labels = O.synth_labels(bs, 10, n_out, seed=40 + 10 * step + rank)
                driver.run_step((audio, labels, None, None), step)
                assert not _ops._deferred
        finally:
            if mode == "side":
                _ops.defer_off_chain = real
                _ops.SIDE_ON_CPU = False
        out[mode] = task.sed_student.arena.flat.detach().clone()
        out[mode + "_parked"] = len(parked)
    flats = [torch.zeros_like(out["side"]) for _ in range(world)]
    dist.all_gather(flats, out["side"])
    if rank == 0:
        torch.save(dict(equal=bool(torch.equal(out["chain"], out["side"])), ranks_equal=bool(torch.equal(flats[0], flats[1])),
                        parked=out["side_parked"]), os.path.join(out_dir, "side.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_side_sections_under_exchange(tmp_path):
    mp.spawn(_side_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    d = torch.load(os.path.join(str(tmp_path), "side.pt"))
    assert d["parked"] >= 2 * (2 + 1 + 6), d          # per step: two BiGRU sections, the head's sums, six CNN weight gradients
    assert d["equal"] and d["ranks_equal"], d


def _rehearsal_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", SED_DDP_REHEARSE="1")
    torch.set_num_threads(1)
    from tests.emu_support import bind_emulator
    bind_emulator()
    import random
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd.launcher import StepDriver, init_distributed
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (0, 1) and dist.is_initialized()
    bs, n_samp = (1, 1, 1), 4096 + 1024
    sd = O.make_state_dict(seed=7)
    audio = O.synth_audio(3, n_samp, seed=100)
    labels = O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5)
    out = {}
    for mode, env, overlap in (("plain", "0", None), ("bucketed", "1", True), ("single", "1", False)):
        os.environ["SED_DDP_REHEARSE"] = env
        task = P.build_task("cpu", bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = StepDriver(task, world_size=1, overlap_allreduce=overlap)
        assert driver.exchange == (mode != "plain") and driver.overlap == (mode == "bucketed")
        random.seed(4); np.random.seed(7); torch.manual_seed(7)
        from desed_task_amd import ops as _ops
        _ops.reseed_dropout()
        for step in range(2):
            driver.run_step((audio.clone(), labels.clone(), None, None), step)
        out[mode] = task.sed_student.arena.flat.detach().clone()
        out[mode + "_log"] = [b[0] for b in driver.bucket_log]
    torch.save(out, os.path.join(out_dir, "rehearsal.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_rank_rehearsal_equals_plain_step(tmp_path):
    """launcher.rehearsing(): the exchange machinery on a one-rank process group (both schemes) changes no bit of two steps."""
    mp.spawn(_rehearsal_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    d = torch.load(os.path.join(str(tmp_path), "rehearsal.pt"))
    assert d["plain_log"] == [] and d["bucketed_log"] == ["A", "B"] and d["single_log"] == ["AB"]
    assert torch.equal(d["plain"], d["bucketed"]) and torch.equal(d["plain"], d["single"])


def test_rank_sharded_batch_sampler():
    from desed_task_amd.launcher import RankShardedBatchSampler

    class Batches:
        def __init__(self):
            self.epoch = None

        def set_epoch(self, e):
            self.epoch = e

        def __len__(self):
            return 7

        def __iter__(self):
            return iter([[10 * i, 10 * i + 1] for i in range(7)])

    base = Batches()
    shards = [RankShardedBatchSampler(base, r, 3) for r in range(3)]
    got = [list(s) for s in shards]
    assert [len(s) for s in shards] == [2, 2, 2] and all(len(g) == 2 for g in got)      # 7 // 3: the tail batch is dropped
    assert got[0] == [[0, 1], [30, 31]] and got[1] == [[10, 11], [40, 41]] and got[2] == [[20, 21], [50, 51]]
    shards[1].set_epoch(5)
    assert base.epoch == 5
    with pytest.raises(ValueError):
        RankShardedBatchSampler(base, 3, 3)

    # the shuffling sampler below draws from torch's global CPU generator, like the recipe's ConcatDatasetBatchSampler
    class Shuffled(Batches):
        def __iter__(self):
            return iter([[int(i)] for i in torch.randperm(7)])

    class FakeCuda:         # torch.manual_seed() reseeds the GPU generators through this hook; the sampler must not call it
        calls = 0

    import torch.cuda as tc
    orig = tc.manual_seed_all
    tc.manual_seed_all = lambda s: setattr(FakeCuda, "calls", FakeCuda.calls + 1)
    try:
        torch.manual_seed(99)
        FakeCuda.calls = 0
        before = torch.get_rng_state().clone()
        a = [RankShardedBatchSampler(Shuffled(), r, 2, seed=3) for r in range(2)]
        torch.manual_seed(1234)                 # rank-local draws in between must not matter
        FakeCuda.calls = 0
        before = torch.get_rng_state().clone()
        e0 = [list(s) for s in a]
        assert torch.equal(torch.get_rng_state(), before), "the caller's CPU generator state must be left untouched"
        assert FakeCuda.calls == 0, "the sampler reseeded the GPU generators"
    finally:
        tc.manual_seed_all = orig
    assert sorted(b[0] for b in e0[0] + e0[1]) == sorted(set(b[0] for b in e0[0] + e0[1])) and len(e0[0]) == len(e0[1]) == 3
    e1 = [list(s) for s in a]                   # no set_epoch() call: the next epoch still gets a new order, same on all ranks
    assert a[0].epoch == a[1].epoch == 2 and e1 != e0


@pytest.mark.timeout(1500)
def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2` from a plain shell (no torchrun environment) starts its own two ranks; here as the dry run on the
    CPU emulator over gloo.  The JSON line must say which backend / world size ran and carry the gradient-bucket log."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=570)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["metric"].startswith("DRY RUN")
    d = out["dist"]
    # default front end: the next step's front half + teacher CNN forward pipelined under backward -> one graph, ONE all-reduce over
    # the whole gradient arena after backward (the two-bucket scheme is asserted in test_two_rank_gradient_average above)
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["overlap_allreduce"] is False
    assert [b[0] for b in d["bucket_log"]] == ["AB"] and abs(d["bucket_log"][0][2] - 4.45) < 0.01
    assert "pipelined" in out["config"]["front_end"]
    assert len(d["ms_per_step_per_rank"]) == 2
    # what makes a future N > 1 line self-explaining: the measured exchange tail, on the host clock here
    tail = d["exchange_tail_us"]
    exposed = [k for k in tail if k.startswith("exposed_exchange")]
    assert exposed and tail[exposed[0]]["host_us"] > 0 and tail[exposed[0]]["device_us"] is None
    assert "backward_done -> exchange_done" in tail and "exchange_done -> adam_done" in tail
    assert "falsified by" in d["expected"] and "rccl_debug" in d
    # the SAME launch at the node's size -- eight ranks (only two had ever been started): rank-local seeds / port /
    # the JSON line from rank 0 only, one all-reduce over the whole arena per step on every rank
    r8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "1"],
                        cwd=ROOT, env=dict(env, SED_EMU_THREADS="2"), capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stderr[-2000:]
    lines = [ln for ln in r8.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r8.stdout[-1000:]
    out8 = json.loads(lines[0])
    d8 = out8["dist"]
    assert out8["n_gpus"] == 8 and d8["world_size"] == 8 and len(d8["ms_per_step_per_rank"]) == 8
    assert out8["config"]["global_batch"] == 8 * out["config"]["global_batch"] // 2 and "cpu_baseline" not in out8
    assert out8["config"]["surface"]["surface"] == "driver"
    # without a GPU and without --dry-run the bench refuses loudly instead of producing a number
    if not torch.cuda.is_available():
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True,
                            text=True, timeout=120)
        assert r2.returncode != 0 and "needs the MI355X" in r2.stderr and not r2.stdout.strip()


REF = "/root/reference"
RECIPE = os.path.join(REF, "recipes", "dcase2023_task4_baseline")


@pytest.mark.timeout(1500)
@pytest.mark.skipif(not os.path.isdir(RECIPE), reason="needs the reference's data pipeline (build container only)")
def test_launcher_cli_trains_two_ranks_and_resumes(tmp_path):
    """SURVEY 8e "Launcher" / VERDICT r05 item 5: `python -m desed_task_amd.launcher --conf_file ... --gpus 2` from a plain shell (no
    torchrun environment) -- here through tests/launcher_emu.py: the same main() on two gloo ranks with the CPU emulator of the kernels bound -- builds the REFERENCE's data sets /
    ConcatDatasetBatchSampler (a miniature DESED; torchaudio.load stubbed) behind RankShardedBatchSampler, trains two epochs, validates on
    the rank-averaged BatchNorm statistics, writes Lightning-shaped checkpoints from rank 0 and tests the best one; the checkpoint loads
    strictly into a fresh reference-shaped SEDTask4 and into torch.optim.Adam; a second command resumes it for one more epoch."""
    import subprocess
    import yaml
    sys.path[:0] = [p for p in (RECIPE, os.path.join(ROOT, "desed_task_amd", "drop_in"), ROOT, REF) if p not in sys.path]
    from tests import mini_desed
    tmp = str(tmp_path)
    stubs = mini_desed.write_stubs(os.path.join(tmp, "stubs"))
    conf = mini_desed.build(tmp, RECIPE, n_epochs=2)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([RECIPE, os.path.join(ROOT, "desed_task_amd", "drop_in"), ROOT, REF, stubs])
    env["SED_EMU_THREADS"] = "3"
    log_dir = os.path.join(tmp, "exp")
    cmd = [sys.executable, "-W", "ignore", "-m", "tests.launcher_emu", "--conf_file", conf, "--log_dir", log_dir, "--gpus", "2"]
    r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = r.stdout
    assert "epoch 0: 2 steps x 2 ranks" in out and "epoch 1: 2 steps x 2 ranks" in out and out.count("val/obj_metric") >= 2 and "best model:" in out
    vdir = os.path.join(log_dir, "version_0")
    files = sorted(os.listdir(vdir))
    best = [f for f in files if f.startswith("epoch=")]
    assert "last.ckpt" in files and len(best) == 1 and os.path.isdir(os.path.join(vdir, "metrics_test")), files
    ckpt = torch.load(os.path.join(vdir, "last.ckpt"), map_location="cpu", weights_only=False)
    assert ckpt["epoch"] == 2 and ckpt["global_step"] == 4 and ckpt["pytorch-lightning_version"].startswith("1.9")
    assert int(ckpt["optimizer_states"][0]["state"][0]["step"]) == 4 and ckpt["lr_schedulers"][0]["step_num"] == 5
    # a fresh SEDTask4 (reference constructor surface) takes the state dict strictly; torch.optim.Adam takes the optimizer state
    from tests.emu_support import bind_emulator
    bind_emulator()
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer import SEDTask4
    hp = ckpt["hyper_parameters"]
    student = CRNN(**hp["net"])
    task = SEDTask4(hp, encoder=None, sed_student=student)
    missing = task.load_state_dict(ckpt["state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    torch.optim.Adam(student.parameters(), 1e-3).load_state_dict(ckpt["optimizer_states"][0])
    w_after_2 = ckpt["state_dict"]["sed_student.cnn.cnn.conv0.weight"].clone()
    # resume: one more epoch on top of last.ckpt
    cfg = yaml.safe_load(open(conf))
    cfg["training"]["n_epochs"] = 3
    conf3 = os.path.join(tmp, "conf3.yaml")
    yaml.safe_dump(cfg, open(conf3, "w"))
    log2 = os.path.join(tmp, "exp_resumed")
    r2 = subprocess.run(cmd[:5] + ["--conf_file", conf3, "--log_dir", log2, "--gpus", "2", "--resume_from_checkpoint",
                                   os.path.join(vdir, "last.ckpt")], cwd=tmp, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    assert "epoch 2: 2 steps x 2 ranks" in r2.stdout and "epoch 0:" not in r2.stdout and "epoch 1:" not in r2.stdout
    ck3 = torch.load(os.path.join(log2, "version_0", "last.ckpt"), map_location="cpu", weights_only=False)
    assert ck3["epoch"] == 3 and ck3["global_step"] == 6 and int(ck3["optimizer_states"][0]["state"][0]["step"]) == 6
    assert not torch.equal(ck3["state_dict"]["sed_student.cnn.cnn.conv0.weight"], w_after_2)


def _worker_whole_step(rank, world, port, out_dir):
    """SEDTask4's whole-step mode under a two-rank process group (SED_DDP_GRAPH_EXCHANGE=1 lifts the blocker): the Lightning-order loop
    (tests/lightning_order.Trainer: module hooks + optimizer.step(closure) only) == launcher.StepDriver(world_size=2) driven by hand."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SED_DDP_GRAPH_EXCHANGE="1")
    torch.set_num_threads(1)
    import random
    from tests.emu_support import bind_emulator, emu_threads
    bind_emulator()
    emu_threads(1)                                              # in-order workgroups: two runs of the same launches give the same bits
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from tests.lightning_order import Trainer
    from desed_task_amd import ops as _ops
    from desed_task_amd.launcher import StepDriver, init_distributed
    from desed_task_amd.lookahead import BatchList
    init_distributed(backend="gloo")
    bs, n_samp, per_epoch = (1, 1, 2), 2048 + 1024, 2
    B, n_out = sum(bs), (1 + n_samp // 256) // 4
    sd = O.make_state_dict(seed=7)
    audios = [O.synth_audio(B, n_samp, seed=300 + 11 * rank + i) for i in range(per_epoch)]     # every rank its own clips
    labelss = [O.synth_labels(bs, 10, n_out, seed=40 + 3 * rank + i) for i in range(per_epoch)]

    class Clips(BatchList):
        def __getitem__(self, i):
            return (audios[i], labelss[i].clone(), [1.0] * B)

    def seed():
        random.seed(41 + rank); np.random.seed(101 + rank); torch.manual_seed(101 + rank)
        _ops.reseed_dropout()

    finals = {}
    for mode in ("whole", "driver"):
        task = P.build_task("cpu", bs, sd, dropout=0.5, specaug=True, rampup=5, torch_adam=mode == "whole", whole_step=mode == "whole",
                            train_data=Clips([None] * per_epoch))
        seed()
        if mode == "whole":
            assert task._whole_step_blockers() is None
            tr = Trainer(max_epochs=2).fit(task)
            assert tr.global_step == 2 * per_epoch and task._driver is not None and task._driver.world == world and task._driver.exchange
            losses = [float(l) for l in tr.losses]
        else:
            driver = StepDriver(task, world_size=world, prefetch="teacher")
            data, losses = Clips([None] * per_epoch), []
            for epoch in range(2):
                batches = list(torch.utils.data.DataLoader(data, batch_size=None))
                for i in range(per_epoch):
                    losses.append(float(driver.run_step(batches[i], i, next_batch=batches[i + 1] if i + 1 < per_epoch else None).detach()))
        finals[mode] = (losses, task.sed_student.arena.flat.detach().clone(), task.sed_teacher.arena.flat.detach().clone(),
                        task.sed_student.cnn.cnn.batchnorm0.running_mean.detach().clone())
    assert finals["whole"][0] == finals["driver"][0], (finals["whole"][0], finals["driver"][0])
    for a, b in zip(finals["whole"][1:], finals["driver"][1:]):
        assert torch.equal(a, b)
    # the students agree across the ranks (their BatchNorm statistics do not: rank-local clips)
    flats = [torch.zeros_like(finals["whole"][1]) for _ in range(world)]
    dist.all_gather(flats, finals["whole"][1])
    assert torch.equal(flats[0], flats[1])
    # without the switch the mode stays blocked under this process group
    os.environ["SED_DDP_GRAPH_EXCHANGE"] = "0"
    assert "SED_DDP_GRAPH_EXCHANGE" in (P.build_task("cpu", bs, sd, torch_adam=True)._whole_step_blockers() or "")
    if rank == 0:
        open(os.path.join(out_dir, "whole_ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1200)
def test_whole_step_mode_under_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker_whole_step, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "whole_ok"))
