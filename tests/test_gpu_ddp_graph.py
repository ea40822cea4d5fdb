"""Data-parallel hipGraph path.  On ONE GPU: two ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks per
device).  On a node with >= 2 GPUs the same worker also runs over RCCL, one rank per device (`test_two_rank_graph_step_rccl`:
skipped on the 1-GPU boxes, runs wherever the driver has a multi-GPU node), incl. the A/B switches of launcher.StepDriver.  Exercises GraphedStepDriver with world_size 2 -- graph replay up to backward,
eager all-reduce of the flat gradient arena, eager Adam -- and checks that the student stays identical across ranks and equals
the eager StepDriver run on the same data."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, overlap, backend="gloo", dw_side="0", prefetch=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SED_DIST_BACKEND=backend, SED_GRU_DW_SIDE=dw_side, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if overlap is None:
        os.environ.pop("SED_DDP_OVERLAP", None)             # the drivers' own default for the chosen front end
    else:
        os.environ["SED_DDP_OVERLAP"] = overlap
    import random
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd.graph import GraphedStepDriver
    from desed_task_amd.launcher import StepDriver, init_distributed
    r, _, w = init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == backend
    if backend == "nccl":
        assert torch.cuda.current_device() == rank          # one rank per device
    dev = "cuda"
    bs, n_samp, steps = (1, 1, 2), 16000 + 1024, 4
    sd = O.make_state_dict(seed=7)
    audio = P.to(dev, O.synth_audio(4, n_samp, seed=100 + rank))            # different clips per rank
    labels = P.to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5 + rank))
    finals = []
    for mode in ("eager", "graph"):
        task = P.build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = StepDriver(task, world, prefetch=prefetch) if mode == "eager" else GraphedStepDriver(task, world, warmup=1, prefetch=prefetch)
        from desed_task_amd import ops as _ops
        for step in range(steps):
            if step == 0 or prefetch is None:       # (pipelined: step k + 1's draws are made during step k -- seed once)
                random.seed(40 + step); np.random.seed(100 + step + 17 * rank); torch.manual_seed(100 + step + 17 * rank)
                torch.cuda.manual_seed(100 + step + 17 * rank)
                _ops.reseed_dropout()
            if prefetch is None:
                driver.run_step((audio.clone(), labels.clone(), None, None), step)
            else:       # the same clips every step; the announced labels are mixed in place one step early
                nxt = (audio, labels.clone(), None, None) if step + 1 < steps else None
                driver.run_step((audio, labels.clone(), None, None), step, next_batch=nxt)
        torch.cuda.synchronize()
        finals.append(task.sed_student.arena.flat.detach().cpu().clone())
    mine = finals[1].to(dev) if backend == "nccl" else finals[1]       # (RCCL moves device tensors only)
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    both = [t.cpu() for t in both]
    split = driver.eager.bucket_bounds()[0]
    two_graphs = driver.graph_cnn is not None
    if rank == 0:
        torch.save(dict(eager=finals[0], graph=finals[1], ranks=both, split=split, two_graphs=two_graphs,
                        dw_side=driver.eager.gru_dw_side, bucket_log=driver.eager.bucket_log), os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_two_rank_graph_step(tmp_path, overlap):
    """overlap 1: bucketed exchange, the step is two graphs with bucket A's all-reduce between them; 0: one graph + one blocking
    all-reduce.  Either way the replayed steps must equal the eager StepDriver on the same draws."""
    _run_and_check(tmp_path, overlap, "gloo", "0")


@pytest.mark.timeout(600)
def test_two_rank_graph_step_pipelined_front_half(tmp_path):
    """Two ranks with the next step's front half + teacher CNN forward pipelined under backward (prefetch "teacher"): the drivers'
    default is then ONE graph and ONE all-reduce over the whole arena; graph == eager, students identical across ranks."""
    _run_and_check(tmp_path, None, "gloo", "0", prefetch="teacher")


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (one RCCL rank per device)")
@pytest.mark.parametrize("overlap,dw_side", [("1", "0"), ("0", "0"), ("1", "1")])
def test_two_rank_graph_step_rccl(tmp_path, overlap, dw_side):
    """The same over RCCL with one rank per GPU: both exchange schemes, and the BiGRU weight-gradient side stream that stays off
    at world > 1 by default (launcher.StepDriver) switched on."""
    d = _run_and_check(tmp_path, overlap, "nccl", dw_side)
    assert d["dw_side"] == (dw_side == "1")


def _rehearsal_worker(rank, port, out_dir, overlap, prefetch, graph_exchange=False):
    """ONE rank over RCCL (launcher.rehearsing): the N > 1 step structure on a one-rank communicator vs the plain one-GPU step."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SED_DDP_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SED_DDP_GRAPH_EXCHANGE="1" if graph_exchange else "0")
    os.environ.pop("SED_DIST_BACKEND", None)
    if overlap is None:
        os.environ.pop("SED_DDP_OVERLAP", None)
    else:
        os.environ["SED_DDP_OVERLAP"] = overlap
    import random
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd import ops as _ops
    from desed_task_amd.graph import GraphedStepDriver
    from desed_task_amd.launcher import init_distributed
    r, _, w = init_distributed()
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    dev = "cuda"
    bs, n_samp, steps = (1, 1, 2), 16000 + 1024, 5
    sd = O.make_state_dict(seed=7)
    audio = [P.to(dev, O.synth_audio(4, n_samp, seed=100 + k)) for k in range(steps)]
    labels = [P.to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5 + k)) for k in range(steps)]
    finals, info = [], {}
    for mode in ("plain", "rehearsal"):
        os.environ["SED_DDP_REHEARSE"] = "1" if mode == "rehearsal" else "0"
        task = P.build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = GraphedStepDriver(task, 1, warmup=1, prefetch=prefetch)
        assert driver.eager.exchange == (mode == "rehearsal")
        random.seed(40); np.random.seed(100); torch.manual_seed(100); torch.cuda.manual_seed(100)
        _ops.reseed_dropout()
        for step in range(steps):
            batch = (audio[step], labels[step].clone(), None, None)
            if prefetch is None:
                driver.run_step(batch, step)
            else:
                nxt = (audio[step + 1], labels[step + 1].clone(), None, None) if step + 1 < steps else None
                driver.run_step(batch, step, next_batch=nxt)
        torch.cuda.synchronize()
        finals.append(torch.cat([task.sed_student.arena.flat.detach().cpu(), task.sed_teacher.arena.flat.detach().cpu()]))
        if mode == "rehearsal":
            info = dict(two_graphs=driver.graph_cnn is not None, bucket_log=driver.eager.bucket_log, overlap=driver.eager.overlap,
                        dw_side=driver.eager.gru_dw_side, capture_exchange=driver.capture_exchange)
    torch.save(dict(plain=finals[0], rehearsal=finals[1], **info), os.path.join(out_dir, "rehearsal.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap,prefetch", [("1", None), ("0", None), (None, "teacher")])
def test_one_rank_rccl_rehearsal_equals_plain_step(tmp_path, overlap, prefetch):
    """The data-parallel step over RCCL on the 1-GPU boxes: a one-rank communicator carries the start-up broadcast, the bucketed
    (two graphs, bucket A asynchronous between them) or single all-reduce and the eager Adam launch.  Sums over one rank change no
    bit, so student AND teacher after 5 steps over 5 different batches must equal the plain one-graph step exactly."""
    mp.spawn(_rehearsal_worker, args=(_free_port(), str(tmp_path), overlap, prefetch), nprocs=1, join=True)
    d = torch.load(os.path.join(str(tmp_path), "rehearsal.pt"))
    want_overlap = (overlap == "1") if overlap is not None else False       # (pipelined "teacher": one graph + one all-reduce)
    # (round 5: with ONE all-reduce after the joined backward pass the weight-gradient GEMMs run beside the chain under RCCL too;
    #  with the bucketed overlap they stay on it -- bucket A is issued from inside the backward pass)
    assert d["overlap"] == want_overlap and d["two_graphs"] == want_overlap and d["dw_side"] is (not want_overlap)
    assert [b[0] for b in d["bucket_log"]] == (["A", "B"] if want_overlap else ["AB"])
    # Strict again (round 4).  Round 3 bounded this comparison after one run of 43 "differed" -- it had not: the worker had DIED (the
    # RCCL watchdog's event query during the stream capture, see graph.quiesce_collectives).  tools/rehearsal_loop.py: 87 repetitions
    # before the fix = 0 numeric differences and 6 dead processes; after it, profiles/r04_rehearsal_loop.md.
    assert torch.equal(d["plain"], d["rehearsal"]), "one-rank rehearsal differs from the plain step: max %.3e" % (
        (d["plain"] - d["rehearsal"]).abs().max().item())


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap,prefetch", [("0", None), ("1", None), (None, "teacher")])
def test_one_rank_rccl_rehearsal_with_the_exchange_captured(tmp_path, overlap, prefetch):
    """SED_DDP_GRAPH_EXCHANGE=1 (round 5): the RCCL all-reduce(s) and Adam are nodes of the ONE captured graph -- no second graph, no
    host hand-over after the replay.  Same strict comparison with the plain one-GPU step over 5 steps (1 eager, the capture, 3 replays)."""
    mp.spawn(_rehearsal_worker, args=(_free_port(), str(tmp_path), overlap, prefetch, True), nprocs=1, join=True)
    d = torch.load(os.path.join(str(tmp_path), "rehearsal.pt"))
    assert d["capture_exchange"] is True and d["two_graphs"] is False
    assert torch.equal(d["plain"], d["rehearsal"]), "captured-exchange rehearsal differs from the plain step: max %.3e" % (
        (d["plain"] - d["rehearsal"]).abs().max().item())


def _b48_rehearsal_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SED_DDP_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("SED_DIST_BACKEND", None)
    os.environ.pop("SED_DDP_OVERLAP", None)
    from tests import parity_cases as P
    from desed_task_amd.launcher import init_distributed
    init_distributed()
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    w = P.case_b48_graph_step_vs_oracle("cuda", prefetch="teacher")
    torch.save(w, os.path.join(out_dir, "b48.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_b48_pipelined_graph_step_vs_oracle_under_rccl_rehearsal(tmp_path):
    """The same B = 48 pipelined-graph oracle comparison through the N > 1 step structure: graph up to the end of backward, the
    RCCL all-reduce over the flat gradient arena (one-rank communicator), eager Adam."""
    mp.spawn(_b48_rehearsal_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    w = torch.load(os.path.join(str(tmp_path), "b48.pt"))
    print("B=48 pipelined graph step under the one-rank RCCL rehearsal, worst errors:", w)
    assert w["modes"] == ["eager", "capture", "replay"] and w["exchange"] is True


def _capture_worker(rank, port, out_dir, reps):
    """Regression test of graph.quiesce_collectives: the step's capture (three side streams forked and joined inside it, ~ 300 launches)
    right behind live RCCL collectives, `reps` times in one process.  On round 3's build ~ 7 % of these captures killed the process --
    ProcessGroupNCCL's watchdog polls the end event of the last collective from its own thread while the main thread captures
    (tools/rehearsal_loop.py: 6 dead processes in 87 repetitions; tools/nodrain_check.py runs THIS worker with the drain disabled: 1 of 4 runs of 30 captures died)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SED_DDP_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("SED_DIST_BACKEND", None)
    os.environ.pop("SED_DDP_OVERLAP", None)
    from oracle import sed_oracle as O
    from tests import parity_cases as P
    from desed_task_amd.graph import GraphedStepDriver
    from desed_task_amd.launcher import init_distributed
    init_distributed()
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    dev = "cuda"
    bs, n_samp = (1, 1, 2), 8192 + 1024
    sd = O.make_state_dict(seed=7)
    audio = P.to(dev, O.synth_audio(4, n_samp, seed=3))
    labels = P.to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5))
    ok = 0
    for i in range(reps):
        task = P.build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
        driver = GraphedStepDriver(task, 1, warmup=1, prefetch="teacher")
        assert driver.eager.exchange
        for step in range(3):                         # eager (all-reduce, eager Adam) -> capture right behind it -> one replay
            driver.run_step((audio, labels, None, None), step, next_batch=(audio, labels, None, None))
        torch.cuda.synchronize()
        assert driver.graph is not None and torch.isfinite(task.sed_student.arena.flat).all()
        ok += 1
        del driver, task
    torch.save(dict(ok=ok), os.path.join(out_dir, "capture.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_capture_next_to_live_rccl_collectives(tmp_path):
    mp.spawn(_capture_worker, args=(_free_port(), str(tmp_path), 60), nprocs=1, join=True)
    assert torch.load(os.path.join(str(tmp_path), "capture.pt"))["ok"] == 60


def _run_and_check(tmp_path, overlap, backend, dw_side, prefetch=None):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), overlap, backend, dw_side, prefetch), nprocs=2, join=True)
    d = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    assert d["two_graphs"] == (overlap == "1")
    assert [b[0] for b in d["bucket_log"]] == (["A", "B"] if overlap == "1" else ["AB"])
    r0, r1 = d["ranks"]
    assert torch.equal(r0, r1)                                     # same reduced gradient + same Adam -> identical students
    diff = (d["eager"] - d["graph"]).abs()
    sp = d["split"]
    msg = "cnn part: max %.2e frac>5e-5 %.3f | tail part: max %.2e frac>5e-5 %.3f" % (
        diff[:sp].max().item(), (diff[:sp] > 5e-5).float().mean().item(), diff[sp:].max().item(), (diff[sp:] > 5e-5).float().mean().item())
    print("two-rank eager vs graph (%s, overlap %s, prefetch %s): %s" % (backend, overlap, prefetch, msg))
    if backend == "nccl":
        # RCCL collectives are stream-ordered: graph == eager bit for bit, like in a single process
        assert torch.equal(d["eager"], d["graph"]), msg
    else:
        # gloo (test backend only: two ranks time-slicing one GPU) stages device tensors through the host on streams of its own; its
        # ordering is fenced with device synchronisations (launcher._gloo_fence) and was bit-exact in 13 of 14 runs at the end of
        # round 3.  The bound stays at Adam's worst case for a sign flip of a near-zero gradient element (2 lr per step) so that a
        # gloo hiccup cannot fail the suite; the difference is printed.
        assert diff.max().item() <= 2.5 * 1e-3 * 4, msg
        assert (diff > 5e-5).float().mean().item() <= 0.05, msg
    return d


@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_the_gpu(tmp_path):
    """`python bench.py --gpus 2` end to end ON THE GPU: self-launch under torch.distributed.run, two ranks, the pipelined hipGraph step
    with the eager exchange tail, the exchange probe with HIP events, one JSON line from rank 0.  The ranks time-slice the one GPU over
    gloo (RCCL refuses two ranks per device), so the numbers mean nothing -- the line says so -- but every line of the N > 1 bench path
    except the collective backend runs on the hardware."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SED_BENCH_SHARE_GPU="1", SED_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "5", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    if torch.cuda.device_count() < 2:          # (on a multi-GPU node the two gloo ranks get a device each: no sharing to flag)
        assert out.get("shared_gpu") is True and out["metric"].startswith("SHARED GPU")
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and len(d["ms_per_step_per_rank"]) == 2
    assert "hipGraph replay" in out["config"]["launch"] and "one graph" in d["graph_scheme"]
    tail = d["exchange_tail_us"]
    exposed = [k for k in tail if k.startswith("exposed_exchange")]
    assert exposed and tail[exposed[0]]["device_us"] is not None and tail[exposed[0]]["device_us"] > 0
    assert out["config"]["global_batch"] == 96 and out["roofline"] is not None


def _whole_step_rehearsal_worker(rank, port, out_dir):
    """SEDTask4's whole-step mode with the gradient exchange CAPTURED (SED_DDP_GRAPH_EXCHANGE=1), rehearsed on a one-rank RCCL group."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SED_DDP_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SED_DDP_GRAPH_EXCHANGE="1")
    os.environ.pop("SED_DIST_BACKEND", None)
    os.environ.pop("SED_DDP_OVERLAP", None)
    from tests import parity_cases as P
    from desed_task_amd import graph as G
    from desed_task_amd.launcher import init_distributed
    init_distributed()
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    seen = []
    orig = G.GraphedStepDriver.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        seen.append((self.eager.exchange, self.capture_exchange))

    G.GraphedStepDriver.__init__ = spy
    # driver by hand == whole-step behind the Lightning-order loop == the hooks one by one (which exchange nothing: sums over one rank
    # change no bit), 2 epochs x 3 batches, bit for bit
    P.case_lightning_surface("cuda", epochs=2, per_epoch=3)
    assert seen and all(s == (True, True) for s in seen), seen
    torch.cuda.synchronize()
    open(os.path.join(out_dir, "whole_rehearsal_ok"), "w").write(str(len(seen)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_whole_step_mode_with_the_exchange_captured_rccl_rehearsal(tmp_path):
    """the whole-step (fast) mode is no longer blocked under a multi-rank process group when the exchange is part of
    the captured step; here the same structure -- all-reduce and Adam as nodes of the ONE graph behind SEDTask4.training_step -- over RCCL
    on one rank (tests/test_ddp_gloo.py::test_whole_step_mode_under_two_ranks runs it on two gloo ranks on the emulator)."""
    mp.spawn(_whole_step_rehearsal_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "whole_rehearsal_ok"))
