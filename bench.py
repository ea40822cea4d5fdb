#!/usr/bin/env python3
"""Throughput of the mel + CRNN mean-teacher TRAINING step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = SEDTask4.training_step (mel -> mixup -> log/min-max -> student CRNN + teacher CRNN, both in train mode
with dropout + SpecAugment -> BCE/MSE losses) + EMA + backward + [gradient all-reduce] + Adam + warm-up scheduler
on one batch of 48 synthetic 10 s / 16 kHz clips per GPU (12 strong / 12 weak / 24 unlabelled), already resident in
HBM.  Weak scaling: every rank runs the reference's single-GPU step on its own clips; gradients are averaged.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel that dominates the step (the 3x3 convolution as
implicit GEMM on the f32 MFMA): algorithmic FLOPs per launch / the mean launch time measured with HIP events inside
the timed region.  `cpu_baseline` times the CPU oracle (oracle/sed_oracle.py, the unfused torch restatement of the
reference step) on this host's cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BATCH = (12, 12, 24)
N_SAMPLES = 160000
N_FRAMES_OUT = 156
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: exact-f32 MFMA = f32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: dense bf16 MFMA peak (the split-bf16 convs issue 3 bf16 MFMAs per product)


def recipe_config():
    return {
        "training": {"batch_size": list(BATCH), "const_max": 2, "num_workers": 0, "ema_factor": 0.999,
                     "self_sup_loss": "mse", "mixup": "soft", "n_epochs_warmup": 50},
        "scaler": {"statistic": "instance", "normtype": "minmax", "dims": [1, 2], "savepath": None},
        "opt": {"lr": 0.001},
        "feats": {"n_mels": 128, "n_filters": 2048, "hop_length": 256, "n_window": 2048, "sample_rate": 16000,
                  "f_min": 0, "f_max": 8000},
        "net": {"dropout": 0.5, "rnn_layers": 2, "n_in_channel": 1, "nclass": 10, "attention": True, "n_RNN_cell": 128,
                "activation": "glu", "rnn_type": "BGRU", "kernel_size": [3] * 7, "padding": [1] * 7, "stride": [1] * 7,
                "nb_filters": [16, 32, 64, 128, 128, 128, 128],
                "pooling": [[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]], "dropout_recurrent": 0,
                "use_embeddings": False},
    }


def synthetic_batch(device, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(N_SAMPLES, dtype=torch.float32) / 16000.0
    audio = 0.1 * torch.randn(sum(BATCH), N_SAMPLES, generator=g)
    f0 = 200.0 + 40.0 * torch.arange(sum(BATCH), dtype=torch.float32).unsqueeze(1)
    audio += 0.3 * torch.sin(2 * np.pi * (f0 * t + 150.0 * t * t)) + 0.2 * torch.sin(2 * np.pi * (3000.0 * t - 120.0 * t * t))
    labels = torch.zeros(sum(BATCH), 10, N_FRAMES_OUT)
    labels[:BATCH[0]] = (torch.rand(BATCH[0], 10, N_FRAMES_OUT, generator=g) < 0.1).float()
    labels[BATCH[0]:BATCH[0] + BATCH[1], :, 0] = (torch.rand(BATCH[1], 10, generator=g) < 0.2).float()
    return audio.to(device), labels.to(device)


class KernelTimer:
    """HIP-event timing of selected C-ABI entry points on torch's current stream (the stream they launch on)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = {}

    def wrap(self, lib):
        orig = lib.call
        timer = self

        def timed_call(name, *args):
            if name not in timer.names:
                return orig(name, *args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(name, *args)
            e1.record()
            key = (name,) + tuple(a for a in args if isinstance(a, int) and 0 < a < 100000 and not isinstance(a, bool))[:6]
            timer.records.setdefault(key, []).append((e0, e1))
        lib.call = timed_call
        self._undo = lambda: setattr(lib, "call", orig)

    def unwrap(self):
        self._undo()

    def summary(self):
        out = {}
        for key, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = (len(ms), float(np.mean(ms)), float(np.sum(ms)))
        return out


def conv_flops(key):
    """Algorithmic FLOPs of one sed_conv3x3 launch: 2 * B*T*F * 9*CIN * COUT (key = (name, B, T, F, CIN, COUT))."""
    _, B, T, F, CIN, COUT = key[:6]
    return 2.0 * B * T * F * 9 * CIN * COUT


def pmc_traffic(cin, cout, F, split):
    """HBM bytes per launch of the matching conv3x3 kernel from the committed rocprofv3 --pmc passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "r01_q_pmc_traffic.json" if split else "r01_pmc_traffic.json")
    try:
        kernels = json.load(open(path))["kernels"]
    except Exception:  # noqa: BLE001
        return None
    tf = min(F, 32)
    for name, rec in kernels.items():
        if name.startswith("%s<%d, %d, %d, true" % ("conv3x3_bf16_kernel" if split else "conv3x3_kernel", cin, cout, tf)):
            return rec["hbm_bytes"]
    return None


def cpu_baseline():
    """One full oracle training step (training_step + EMA + backward + Adam) on a 12-clip batch of 10 s clips."""
    from oracle import sed_oracle as O
    threads = min(os.cpu_count() or 1, 32)      # torch CPU convs stop scaling (and can collapse) far below 256 threads
    torch.set_num_threads(threads)
    bs = (2, 2, 4)
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn(B, N_SAMPLES, generator=g)
    labels = O.synth_labels(bs, 10, N_FRAMES_OUT, seed=5)
    tr = O.OracleTrainer(sd, batch_sizes=bs)

    def draws():
        mix = dict(c_weak=float(np.random.beta(0.2, 0.2)), perm_weak=torch.randperm(bs[1]),
                   c_strong=float(np.random.beta(0.2, 0.2)), perm_strong=torch.randperm(bs[0]))
        augs, drops = [], []
        for _ in range(2):
            fb = O.specaug_bounds(torch.rand(B), torch.rand(B), 10, 128)
            tb = O.specaug_bounds(torch.rand(B), torch.rand(B), 5, 626)
            augs.append(dict(f=fb, t=tb))
            shapes = [(B, 16, 626, 128), (B, 32, 313, 64), (B, 64, 156, 32), (B, 128, 156, 16), (B, 128, 156, 8),
                      (B, 128, 156, 4), (B, 128, 156, 2), (B, 156, 256)]
            drops.append([(torch.rand(s) >= 0.5).float() for s in shapes])
        return mix, augs, drops

    # tiny warm-up (thread pool, allocator), not timed
    warm = O.OracleTrainer(sd, batch_sizes=(1, 1, 2))
    tot, _ = warm.training_step(audio[:4, :16000], O.synth_labels((1, 1, 2), 10, 15, seed=1))
    warm.optimizer_step(tot)
    t0 = time.perf_counter()
    mix, augs, drops = draws()
    tot, _ = tr.training_step(audio, labels, mix=mix, aug_s=augs[0], aug_t=augs[1], drop_s=drops[0], drop_t=drops[1])
    tr.optimizer_step(tot)
    dt = time.perf_counter() - t0
    return {"value": B / dt, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "1 full oracle training step (mel+mixup+student/teacher fwd+losses+EMA+bwd+Adam), batch %d (%d/%d/%d) of "
                      "10 s clips, dropout+SpecAugment on, fp32 torch CPU, %d threads of %d host cores, %.2f s wall"
                      % (B, bs[0], bs[1], bs[2], threads, os.cpu_count() or 1, dt)}


def cpu_baseline_bounded(timeout_s=150):
    """Run cpu_baseline() in a child process with a hard wall-clock bound (a pathological host must not stall the bench)."""
    import subprocess
    code = "import json, bench; print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline()))"
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: " + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "oracle step on 8 clips did not finish within %d s on this host" % timeout_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying the captured hipGraph step")
    ap.add_argument("--embeddings", action="store_true",
                    help="secondary workload (SURVEY 8f rank 3): the 2023 'pretrained' step, frozen BEATs-shaped embeddings "
                         "(768 x 496 per clip) fused into the CRNN (confs/pretrained.yaml); not the headline metric")
    args = ap.parse_args()

    from desed_task_amd import _lib
    from desed_task_amd.arena import FusedAdam
    from desed_task_amd.launcher import StepDriver, init_distributed
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer import SEDTask4
    from desed_task_amd.utils.schedulers import ExponentialWarmup

    rank, local, world = init_distributed()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    dev = torch.device("cuda", local)
    torch.manual_seed(1234 + rank); np.random.seed(1234 + rank); random.seed(1234 + rank)

    config = recipe_config()
    if args.embeddings:
        from desed_task_amd.sed_trainer_pretrained import SEDTask4      # noqa: F811
        config["net"].update(use_embeddings=True, embedding_size=768, embedding_type="frame", aggregation_type="pool1d")
        config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
    student = CRNN(**config["net"]).to(dev)
    if world > 1:                                   # identical initial weights on every rank
        dist.broadcast(student.arena.flat, src=0)
    opt = FusedAdam(student.parameters(), lr=1e-3, betas=(0.9, 0.999), arena=student.arena)
    sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 50 * 118), "interval": "step"}

    class Enc:
        labels = list(range(10))
    task = SEDTask4(config, Enc(), student, opt=opt, scheduler=sched).to(dev)
    opt.arena = task.sed_student.arena
    task.train()
    if os.environ.get("SED_OVERLAP_TAILS") is not None:
        task.overlap_tails = os.environ["SED_OVERLAP_TAILS"] == "1"
    use_graph = not args.no_graph
    if use_graph:
        # the step is captured once into a hipGraph (desed_task_amd/graph.py) and replayed: 3 eager steps, 1 capture step
        from desed_task_amd.graph import GraphedStepDriver
        driver = GraphedStepDriver(task, world_size=world, warmup=3)
    else:
        driver = StepDriver(task, world_size=world)
    audio, labels = synthetic_batch(dev, 1234 + rank)
    emb = None
    if args.embeddings:
        emb = torch.randn(sum(BATCH), 768, 496, device=dev, generator=torch.Generator(device=dev).manual_seed(77 + rank))

    inputs = {"audio": audio, "emb": emb}

    def one_step(i):
        # graph mode: the driver copies every batch tensor into its static input buffers, so `labels` (mixed in place by the
        # step) needs no clone; eager mode works on the tensors it is given
        captured = use_graph and getattr(driver, "graph", None) is not None
        driver.run_step((inputs["audio"], labels if captured else labels.clone(), None, inputs["emb"]), i)

    # untimed: the W warm-up steps, plus (graph mode) whatever is still missing for the capture to lie outside the timed region
    n_untimed = max(args.warmup, 5) if use_graph else args.warmup
    graph_note = None
    for i in range(n_untimed):
        try:
            one_step(i)
        except RuntimeError as exc:             # a capture the runtime refuses must not cost the measurement: eager launches
            if not use_graph or driver.graph is not None and driver.n > driver.warmup + 1:
                raise
            graph_note = "hipGraph capture failed (%s): eager launches" % str(exc).splitlines()[0][:120]
            sys.stderr.write(graph_note + "\n")
            use_graph = False
            driver = driver.eager
            one_step(i)
    if use_graph and driver.input_buffers() is not None:
        # the synthetic clips already live in HBM: hand the graph's own input buffers back as the batch, like a loader that
        # writes its batches straight into them, so that no per-step staging copy of the 30 MB of audio is timed.  The labels are
        # mixed in place by the step and are therefore re-staged every step.
        bufs = driver.input_buffers()
        inputs["audio"] = bufs[0]
        if emb is not None:
            inputs["emb"] = bufs[3]
    timer = KernelTimer({"sed_conv3x3", "sed_conv3x3_bf16x3"})
    if not use_graph:
        timer.wrap(_lib.get())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_graph:
        # per-launch HIP events cannot be placed inside a graph replay: time the roofline kernel over eager steps of the
        # same workload right after the timed region (same process, same tensors, same stream)
        timer.wrap(_lib.get())
        for i in range(5):
            driver.eager.run_step((audio, labels.clone(), None, emb), i)     # (eager: works on the tensors it is given)
        torch.cuda.synchronize()
    timer.unwrap()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(task.logged["train/student/loss_strong"])
    if rank != 0:
        return

    summ = timer.summary()
    dom_key, dom = None, None
    for key, (n, mean_ms, tot_ms) in summ.items():
        if dom is None or tot_ms > dom[2]:
            dom_key, dom = key, (n, mean_ms, tot_ms)
    roofline = None
    if dom_key is not None:
        fl = conv_flops(dom_key)
        achieved = fl / (dom[1] * 1e-3) / 1e12
        split = dom_key[0] == "sed_conv3x3_bf16x3"
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": "%s<CIN=%d,COUT=%d> (B,T,F)=(%d,%d,%d), %s" %
                              ("conv3x3_bf16_kernel" if split else "conv3x3_kernel", dom_key[4], dom_key[5], dom_key[1], dom_key[2],
                               dom_key[3], "split-bf16: 3 x v_mfma_f32_32x32x16_bf16 per product, fp32-level accuracy" if split
                               else "v_mfma_f32_32x32x2_f32"),
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "mfma_issue_frac": round((3.0 if split else 1.0) * achieved / peak, 4),
                    "traffic": pmc_traffic(dom_key[4], dom_key[5], dom_key[3], split),
                    "traffic_note": "HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes on this "
                                    "kernel (profiles/r01_q_pmc_fetch_write.md, r01_q_pmc_traffic.json); algorithmic in+out+weights = %d "
                                    "bytes" % (4 * dom_key[1] * dom_key[2] * dom_key[3] * (dom_key[4] + dom_key[5]) + 36 * dom_key[4] * dom_key[5]),
                    "note": "achieved = algorithmic FLOPs (2*B*T*F*9*CIN*COUT) / mean launch time (HIP events; %s); for the "
                            "split-bf16 kernel the MFMA pipe issues 3x that (mfma_issue_frac); f32-equivalent peak would be %.1f"
                            % ("5 eager steps right after the timed region, whose steps are hipGraph replays" if use_graph
                               else "timed region", PEAK_F32_MFMA_TFLOPS),
                    "launches_timed": dom[0], "avg_launch_ms": round(dom[1], 4),
                    "algorithmic_gflop_per_launch": round(fl / 1e9, 3),
                    "conv_share_of_step": round(sum(v[2] for v in summ.values()) / (5 if use_graph else args.steps)
                                                    / (dt * 1e3 / args.steps), 3)}
    clips = sum(BATCH) * world * args.steps
    out = {
        "metric": "10s-clips/sec CRNN mean-teacher train @batch48",
        "value": round(clips / dt, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (fp32 storage and accumulation; the dense contractions of the wide layers run as split-bf16 MFMA, 3 bf16 "
                 "products per fp32 product, fp32-level accuracy; everything else exact f32)", "data": "synthetic",
        "config": {"workload": "dcase2023 CRNN mean-teacher train step, 128-mel 10s@16kHz, batch 48/GPU (12 strong/12 weak/24 "
                               "unlabelled), dropout+SpecAugment+mixup on, fp32 accuracy (conv_precision=%s)%s"
                               % (task.sed_student.cnn.conv_precision,
                                  " + frozen 768x496 embeddings per clip fused by pool1d/cat_tf (confs/pretrained.yaml; secondary "
                                  "workload, SURVEY 8f rank 3)" if args.embeddings else ""),
                   "global_batch": sum(BATCH) * world, "parallelism": "dp%d" % world, "last_loss_strong": round(loss_val, 5),
                   "launch": "hipGraph replay of the captured step (3 eager + 1 capture step before the timed region)" if use_graph
                             else (graph_note or "eager launches"), "untimed_steps": n_untimed},
        "roofline": roofline,
        # SURVEY 8(d) step-level yardsticks (algorithmic work per clip x measured clips/s, per GPU)
        "step_roofline": {
            "mfma_tflops": round(6.464e9 * clips / dt / world / 1e12, 2),
            "mfma_frac_of_f32_peak": round(6.464e9 * clips / dt / world / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            "mfma_frac_of_bf16_peak": round(6.464e9 * clips / dt / world / (PEAK_BF16_MFMA_TFLOPS * 1e12), 5),
            "mel_hbm_frac": round(960512.0 * clips / dt / world / 8.0e12, 6),
            "note": "6.464 GFLOP/clip = conv1-6 + GLU1-6, student fwd + dgrad + wgrad + teacher fwd; 960 512 B/clip = mel path in+out; "
                    "whole-step clips/s, so both are diluted by the GRU recurrence and the HBM-bound narrow blocks (DESIGN.md 8)"},
    }
    if world == 1 and not args.no_cpu_baseline and not args.embeddings:
        out["cpu_baseline"] = cpu_baseline_bounded()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
