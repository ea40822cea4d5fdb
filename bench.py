#!/usr/bin/env python3
"""Throughput of the mel + CRNN mean-teacher TRAINING step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` from a plain shell (no torchrun environment) starts its own N ranks: it re-executes itself under
`torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one process per GPU, rank r -> device r, RCCL).
The JSON line then carries a `dist` object: backend, world size, the gradient-bucket log of the last step, which graph scheme
ran, and the per-rank step times.  A/B switches for a real node: --no-overlap (= SED_DDP_OVERLAP=0: one blocking all-reduce,
one graph), --gru-dw-side (= SED_GRU_DW_SIDE=1: force the BiGRU / CNN weight-gradient GEMMs beside the backward chain at world > 1;
unset they are there over RCCL whenever the gradients leave as ONE all-reduce, SED_GRU_DW_SIDE=0 puts them back on the chain), --no-graph.
`--dry-run` (no GPU needed): the same program on the CPU fiber emulator of the kernels over gloo, at toy sizes -- a plumbing
check of the launch path only; its numbers mean nothing and the line says so.

One "step" = SEDTask4.training_step (mel -> mixup -> log/min-max -> student CRNN + teacher CRNN, both in train mode
with dropout + SpecAugment -> BCE/MSE losses) + EMA + backward + [gradient all-reduce] + Adam + warm-up scheduler
on one batch of 48 synthetic 10 s / 16 kHz clips per GPU (12 strong / 12 weak / 24 unlabelled), already resident in
HBM.  Weak scaling: every rank runs the reference's single-GPU step on its own clips; gradients are averaged.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel that dominates the step -- the launch shape with the largest
total time per step over ALL C-ABI entry points, every launch bracketed by HIP events on its stream over 5 eager steps right
after the timed region -- as algorithmic work per launch / mean launch time against the roof that bounds it;
`roofline_families` is the same accounting per kernel family (mel, block0, conv, glu, wgrad, gru, gemm, head+loss, optim).
`cpu_baseline` times the CPU oracle (oracle/sed_oracle.py, the unfused torch restatement of the reference step) on this
host's cores at the same batch of 48 clips, rank 0, N=1 only.
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
    """HIP-event timing of C-ABI entry points on the stream they launch on (torch's current stream at the call: every
    ops.py / features.py wrapper passes exactly that stream to the library).  names = None: every entry point."""

    def __init__(self, names=None):
        self.names = None if names is None else set(names)
        self.records = {}

    def wrap(self, lib):
        orig = lib.call
        timer = self

        def timed_call(name, *args):
            if timer.names is not None and name not in timer.names:
                return orig(name, *args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(name, *args)
            e1.record()
            nd = ENTRIES[name][2] if name in ENTRIES else 0        # leading shape arguments (pointers are > 2^31, flags come later)
            key = (name,) + tuple(a for a in args if isinstance(a, int) and 0 < a < 2 ** 31 and not isinstance(a, bool))[:nd]
            timer.records.setdefault(key, []).append((e0, e1))
        lib.call = timed_call
        self._undo = lambda: setattr(lib, "call", orig)

    def unwrap(self):
        self._undo()

    def summary(self):
        """key -> (launches, MEDIAN ms per launch, launches x median).  Medians, not means: on a fresh box one launch inside the
        eager window can stall for milliseconds (round 2's driver line had 16 ms / step of "other" from one such stall)."""
        out = {}
        for key, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            med = float(np.median(ms))
            out[key] = (len(ms), med, med * len(ms))
        return out


GLU_BWD128_SPLIT = True           # the split-bf16 16x16x32 kernel is the default 128-channel GLU backward
PEAK_HBM_GBS = 8000.0             # same guide: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)

# entry point -> (kernel that does the work, family, number of leading shape arguments, algorithmic work of ONE call).
# key = (name, the leading shape arguments in call order); work in FLOP for mfma / valu, in bytes for hbm.
# "valu": the recurrence runs on the f32 vector pipes, whose peak equals the exact-f32 MFMA peak (157.3 TFLOP/s).
def _conv_fl(k): return 2.0 * k[1] * k[2] * k[3] * 9 * k[4] * k[5]            # (B,T,F,CIN,COUT)
def _glu_shape(k): return k[1] * k[2] * k[3], k[4], k[5] * k[6]               # (B,T,F,C,PT,PF) -> pixels, C, pool
def _gemm_fl(k): return 2.0 * k[1] * k[2] * k[3]                              # (M,N,K)


def _glu_fwd_work(k):
    npx, C, pool = _glu_shape(k)
    return ("hbm", 4.0 * npx * C * (1 + 1.0 / pool)) if C <= 32 else ("mfma", 2.0 * npx * C * C)


def _glu_bwd_work(k):
    npx, C, pool = _glu_shape(k)
    return ("hbm", 4.0 * npx * C * (2 + 1.0 / pool)) if C <= 32 else ("mfma", 6.0 * npx * C * C)


ENTRIES = {
    "sed_mel_fwd": ("mel_kernel", "mel", 6, lambda k: ("hbm", 4.0 * k[1] * (k[2] + k[3] * k[6]))),      # (B,N,T,n_fft,hop,n_mels)
    "sed_mel_fwd_wave": ("mel_wave_kernel", "mel", 6, lambda k: ("hbm", 4.0 * k[1] * (k[2] + k[3] * k[6]))),
    "sed_logscale_fwd": ("minmax_partial/apply_kernel", "mel", 2, lambda k: ("hbm", 8.0 * k[1] * k[2])),
    "sed_conv0_fwd": ("conv0_kernel", "block0", 4, lambda k: ("hbm", 4.0 * k[1] * k[2] * k[3] * (1 + k[4]))),
    "sed_conv0_wgrad": ("conv0_wgrad_kernel", "block0", 4, lambda k: ("hbm", 4.0 * k[1] * k[2] * k[3] * (1 + 2 * k[4]))),
    # fused first block (B,T,F): reads x, writes the pooled 16-channel output / reads x and the pooled gradient (VALU-bound by design:
    # the conv is recomputed instead of read; the HBM figure is the algorithmic traffic)
    "sed_block0_fwd": ("block0_fwd_kernel", "block0", 3, lambda k: ("hbm", 4.0 * k[1] * k[2] * k[3] * (1 + 16 / 4))),
    "sed_block0_bwd": ("block0_bwd_kernel", "block0", 3, lambda k: ("hbm", 4.0 * k[1] * k[2] * k[3] * (1 + 16 / 4))),
    "sed_conv3x3": ("conv3x3_kernel", "conv", 5, lambda k: ("mfma", _conv_fl(k))),
    "sed_conv3x3_bf16x3": ("conv3x3_bf16_kernel", "conv", 5, lambda k: ("mfma", _conv_fl(k))),
    # (dz, ybn, stats, gamma, dgamma, dbeta, Wd, dx, dy, dbias | B,T,F,CIN,COUT): data gradient with the BatchNorm backward folded in
    "sed_conv3x3_bf16x3_bnbwd": ("conv3x3_bf16_kernel", "conv", 5, lambda k: ("mfma", _conv_fl(k))),
    "sed_conv_wgrad": ("conv_wgrad_kernel", "wgrad", 5, lambda k: ("mfma", _conv_fl(k))),
    "sed_conv_wgrad_bf16x3": ("conv_wgrad_bf16_kernel", "wgrad", 5, lambda k: ("mfma", _conv_fl(k))),
    "sed_glu_fwd": ("glu*_fwd_kernel", "glu", 6, _glu_fwd_work),
    "sed_glu_bwd": ("glu*_bwd_kernel", "glu", 6, _glu_bwd_work),
    "sed_bn_bwd_apply": ("bn_bwd_apply_kernel", "glu", 2, lambda k: ("hbm", 12.0 * k[1] * k[2])),              # (npix, C)
    "sed_gru_fwd": ("gru_fwd_kernel", "gru", 3, lambda k: ("valu", 2.0 * k[1] * k[2] * 2 * 3 * k[3] * k[3])),          # (B,T,H)
    "sed_gru_bwd": ("gru_bwd_kernel", "gru", 3, lambda k: ("valu", 2.0 * k[1] * k[2] * 2 * 3 * k[3] * k[3])),
    "sed_gemm": ("gemm_vec_kernel", "gemm", 3, lambda k: ("mfma", _gemm_fl(k))),
    "sed_gemm_bf16x3": ("gemm_bf16x3_kernel", "gemm", 3, lambda k: ("mfma", _gemm_fl(k))),
    "sed_gemm_pair": ("gemm_vec_kernel", "gemm", 3, lambda k: ("mfma", 2 * _gemm_fl(k))),
    # (M, N, K) per direction.  K <= 256 = the BiGRU input projections (gi = x . [W_f ; W_r]^T + b: 23 MB written for 0.4 - 0.8 MB of
    # weights): bounded by the write stream, priced against HBM -- read x once, both weight matrices, write both outputs
    "sed_gemm_pair_bf16x3": ("gemm_bf16x3_kernel", "gemm", 3,
                             lambda k: ("hbm", 4.0 * (k[1] * k[3] + 2 * k[2] * k[3] + 2 * k[1] * k[2])) if k[3] <= 256 else ("mfma", 2 * _gemm_fl(k))),
    "sed_gemm_pair_splitk_bf16x3": ("gemm_bf16x3_kernel", "gemm", 3, lambda k: ("mfma", 2 * _gemm_fl(k))),
    "sed_gemm_kcat": ("gemm_vec_kernel", "gemm", 3, lambda k: ("mfma", _gemm_fl(k))),
    "sed_gemm_kcat_bf16x3": ("gemm_bf16x3_kernel", "gemm", 3, lambda k: ("mfma", _gemm_fl(k))),
    "sed_head_fwd": ("head_fwd_kernel", "head+loss", 4, lambda k: ("hbm", 4.0 * k[1] * k[2] * (k[3] + 2 * k[4]))),   # (B,T,D,NC)
    "sed_head_bwd": ("head_bwd_kernel", "head+loss", 4, lambda k: ("hbm", 4.0 * k[1] * k[2] * (2 * k[3] + 4 * k[4]))),
    "sed_mt_loss": ("loss_kernel", "head+loss", 3, lambda k: ("hbm", 4.0 * k[1] * k[2] * k[3] * 4)),
    "sed_adam_step": ("adam_kernel", "optim", 1, lambda k: ("hbm", 28.0 * k[1])),
    "sed_ema_update": ("ema_kernel", "optim", 1, lambda k: ("hbm", 12.0 * k[1])),
}


def entry_peak(name, bound, precision, key=()):
    if bound == "hbm":
        return PEAK_HBM_GBS, "GB/s"
    if bound == "valu":
        return PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
    split = "bf16x3" in name or (name in ("sed_glu_fwd", "sed_glu_bwd") and precision == "bf16x3")
    if name == "sed_glu_bwd" and len(key) > 4 and key[4] == 128 and not GLU_BWD128_SPLIT:
        split = False                    # the 128-channel GLU backward runs on the exact-f32 MFMA
    return (PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS), "TFLOP/s"


def roofline_tables(summ, n_steps, step_ms, precision):
    """Per-launch-shape rows and per-family totals from the event timings of `n_steps` eager steps."""
    rows, fam = [], {}
    for key, (n, mean_ms, tot_ms) in summ.items():
        name = key[0]
        ent = ENTRIES.get(name)
        per_step_us = tot_ms * 1e3 / n_steps
        if ent is None:
            f = fam.setdefault("other", {"us_per_step": 0.0})
            f["us_per_step"] += per_step_us
            continue
        kernel, family, _nd, work_fn = ent
        try:
            bound, work = work_fn(key)
        except IndexError:
            bound, work = "hbm", 0.0
        peak, unit = entry_peak(name, bound, precision, key)
        achieved = work / (mean_ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
        rows.append({"entry": name, "kernel": kernel, "family": family, "shape": list(key[1:]), "bound": bound,
                     "launches_per_step": round(n / n_steps, 2), "avg_us": round(mean_ms * 1e3, 2),
                     "us_per_step": round(per_step_us, 1), "work": work, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                     "frac": round(achieved / peak, 4)})
        f = fam.setdefault(family, {"us_per_step": 0.0, "work": {}})
        f["us_per_step"] += per_step_us
        f["work"].setdefault((bound, peak, unit), [0.0, 0.0])
        f["work"][(bound, peak, unit)][0] += work * n / n_steps
        f["work"][(bound, peak, unit)][1] += per_step_us
    families = {}
    for name, f in fam.items():
        e = {"us_per_step": round(f["us_per_step"], 1), "share_of_step": round(f["us_per_step"] / (step_ms * 1e3), 4)}
        for (bound, peak, unit), (work, us) in f.get("work", {}).items():
            ach = work / (us * 1e-6) / (1e9 if bound == "hbm" else 1e12) if us > 0 else 0.0
            e[bound] = {"achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "us_per_step": round(us, 1)}
        families[name] = e
    rows.sort(key=lambda r: -r["us_per_step"])
    return rows, families


def kernel_name(row):
    """rocprof kernel name (prefix) of a roofline row, for the PMC lookup."""
    e, k = row["entry"], row["shape"]
    if e in ("sed_conv3x3", "sed_conv3x3_bf16x3", "sed_conv_wgrad", "sed_conv_wgrad_bf16x3") and len(k) >= 5:
        return "%s<%d, %d" % (row["kernel"], k[3], k[4])
    return row["kernel"]


def lib_rev():
    from desed_task_amd.build import source_rev
    return source_rev()


def _traffic_files():
    """Committed rocprofv3 --pmc traffic summaries, newest first, that were taken on THIS build of the library (same `lib_rev`).
    A summary of another build says nothing about the kernels timed here: it is skipped, and the caller reports null."""
    import glob
    rev = lib_rev()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), key=os.path.getmtime, reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if rec.get("lib_rev") == rev:
            yield path, rec


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` (name prefix) from the newest committed rocprofv3 --pmc summary of this build, or None."""
    for path, rec in _traffic_files():
        kernels = rec["kernels"]
        hits = [rec["hbm_bytes"] for name, rec in kernels.items() if name and name.startswith(kernel.split("*")[0])]
        if hits:
            return max(hits)
    return None


def pmc_step_traffic():
    """(HBM bytes per training step, file) = sum over all kernels of PMC bytes per launch x launches / steps, from the newest
    committed profiles/*_pmc_traffic.json that carries it (tools/pmc_traffic_json.py with the step count), or (None, None)."""
    for path, rec in _traffic_files():
        if rec.get("hbm_bytes_per_step"):
            return int(rec["hbm_bytes_per_step"]), os.path.basename(path)
    return None, None


def cpu_baseline(threads, warmup=2, timed=5, budget_s=120.0):
    """The oracle's full training step (training_step + EMA + backward + Adam) at B = 48 (12/12/24 clips of 10 s, dropout +
    SpecAugment + mixup on), fp32 torch CPU, at `threads` threads: `warmup` untimed + up to `timed` timed steps (fewer if the time
    budget runs out), median.  Dropout keep-masks are drawn outside the timed region (the reference draws them inside: this
    baseline is, if anything, faster than the reference)."""
    from oracle import sed_oracle as O
    torch.set_num_threads(threads)
    bs = BATCH
    B = sum(bs)
    sd = O.make_state_dict(seed=7)
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn(B, N_SAMPLES, generator=g)
    labels = O.synth_labels(bs, 10, N_FRAMES_OUT, seed=5)
    tr = O.OracleTrainer(sd, batch_sizes=bs)
    shapes = [(B, 16, 626, 128), (B, 32, 313, 64), (B, 64, 156, 32), (B, 128, 156, 16), (B, 128, 156, 8),
              (B, 128, 156, 4), (B, 128, 156, 2), (B, 156, 256)]

    def draws():
        mix = dict(c_weak=float(np.random.beta(0.2, 0.2)), perm_weak=torch.randperm(bs[1]),
                   c_strong=float(np.random.beta(0.2, 0.2)), perm_strong=torch.randperm(bs[0]))
        augs, drops = [], []
        for _ in range(2):
            fb = O.specaug_bounds(torch.rand(B), torch.rand(B), 10, 128)
            tb = O.specaug_bounds(torch.rand(B), torch.rand(B), 5, 626)
            augs.append(dict(f=fb, t=tb))
            drops.append([(torch.rand(s) >= 0.5).float() for s in shapes])
        return mix, augs, drops

    def one_step():
        mix, augs, drops = draws()
        t0 = time.perf_counter()
        tot, _ = tr.training_step(audio, labels, mix=mix, aug_s=augs[0], aug_t=augs[1], drop_s=drops[0], drop_t=drops[1])
        tr.optimizer_step(tot)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    warm = [one_step() for _ in range(warmup)]
    times = []
    while len(times) < timed and (not times or time.perf_counter() - t_start + max(times) < budget_s):
        times.append(one_step())
    return {"threads": threads, "clips_per_s": B / float(np.median(times)), "median_s": float(np.median(times)), "min_s": min(times),
            "max_s": max(times), "timed": len(times), "warmup_s": warm}


def _cpu_child(threads, warmup, timed, budget_s, timeout_s):
    import subprocess
    code = ("import json, bench; print('CPU_BASELINE ' + json.dumps(bench.cpu_baseline(%d, %d, %d, %f)))"
            % (threads, warmup, timed, budget_s))
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"threads": threads, "error": "failed: " + r.stderr[-160:]}
    except subprocess.TimeoutExpired:
        return {"threads": threads, "error": "did not finish %d warm-up + 1 timed step within %d s" % (warmup, timeout_s)}


def cpu_baseline_bounded():
    """SURVEY 8(d) / BASELINE.md 3: B = 48, 3 warm-up + 5 timed steps, median -- at the thread count that is fastest on this
    host.  torch's CPU convolutions stop scaling far below a 256-core host's core count (and a step at os.cpu_count() threads on a
    loaded host can take minutes), so every thread count runs in its own child process under a hard wall-clock bound: 32 threads
    (3 + 5 steps), then 64 and os.cpu_count() threads (1 + 2 steps each, 60 s); the fastest median is `value`, all are reported."""
    ncpu = os.cpu_count() or 1
    runs = [_cpu_child(min(32, ncpu), 3, 5, 120.0, 200)]
    for threads in sorted({min(64, ncpu), ncpu} - {min(32, ncpu)}):
        runs.append(_cpu_child(threads, 1, 2, 40.0, 60))
    ok = [r for r in runs if "clips_per_s" in r]
    desc = "; ".join("%d threads: %s" % (r["threads"], ("median %.2f s/step over %d timed steps (min %.2f, max %.2f) = %.2f clips/s"
                                                      % (r["median_s"], r["timed"], r["min_s"], r["max_s"], r["clips_per_s"]))
                                         if "clips_per_s" in r else r["error"]) for r in runs)
    head = ("oracle (unfused fp32 torch-CPU restatement of the reference step: mel+mixup+student/teacher fwd+losses+EMA+bwd+Adam), "
            "batch 48 (12/12/24) of 10 s clips, dropout+SpecAugment+mixup on, host has %d cores; " % ncpu)
    if not ok:
        return {"value": None, "unit": "clips/s", "cores": ncpu, "kind": "port", "sample": head + desc}
    best = max(ok, key=lambda r: r["clips_per_s"])
    return {"value": round(best["clips_per_s"], 3), "unit": "clips/s", "cores": best["threads"], "kind": "port", "sample": head + desc}


def rccl_debug_lines(limit=12):
    """With NCCL_DEBUG=INFO: what RCCL itself says it runs -- the lines of this rank's debug file that name an algorithm / protocol
    / ring or tree (main() points NCCL_DEBUG_FILE at /tmp/sed_bench_rccl_<pid>.log).  None when RCCL debugging is off."""
    path = os.environ.get("NCCL_DEBUG_FILE")
    if not path or os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        return None
    path = path.replace("%p", str(os.getpid())).replace("%h", os.uname().nodename)
    try:
        with open(path, errors="replace") as fh:
            lines = [ln.strip() for ln in fh if any(k in ln for k in ("Algo", "proto", "Protocol", "Ring", "Tree", "Channel", "AllReduce"))]
    except OSError:
        return ["(no RCCL debug file at %s)" % path]
    seen, out = set(), []
    for ln in lines:
        key = ln.split("NCCL INFO", 1)[-1].strip()
        if key not in seen:
            seen.add(key)
            out.append(key[:200])
    return out[-limit:]


def SHARE_GPU():
    """SED_BENCH_SHARE_GPU=1 with SED_DIST_BACKEND=gloo: several ranks time-slice the visible GPU(s) (RCCL refuses two ranks per device).
    A LAUNCH-PATH check of the N > 1 bench on a 1-GPU box (tests/test_gpu_ddp_graph.py) -- the JSON line then says so and is no measurement."""
    return os.environ.get("SED_BENCH_SHARE_GPU") == "1" and os.environ.get("SED_DIST_BACKEND") == "gloo"


def free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def self_launch(n, dry_run):
    """`python bench.py --gpus N` without a torchrun environment: re-execute this command line under torch.distributed.run, one
    process per GPU on this node (rank r binds device r in launcher.init_distributed), rendezvous on 127.0.0.1 at a free port.
    Rank 0's JSON line passes through on stdout; returns the launcher's exit code."""
    import subprocess
    if not dry_run:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs the MI355X (no GPU visible); `--dry-run` checks the launch path on the CPU emulator")
        if n > torch.cuda.device_count() and not SHARE_GPU():
            raise SystemExit("--gpus %d but only %d GPU(s) visible" % (n, torch.cuda.device_count()))
    port = free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's intra-node transport needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def main():
    # The JSON line must be the LAST thing on this job's stdout.  RCCL writes a version banner ("RCCL version : ... Librccl path : ...")
    # to stdout through C stdio when NCCL_DEBUG asks for it, and a piped C buffer is only flushed at exit -- i.e. AFTER Python's print.
    # Ranks other than 0 therefore send their whole stdout to stderr, and rank 0 flushes the C buffers before it prints.
    if int(os.environ.get("RANK", "0")) != 0:
        os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # >= 50 timed steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying the captured hipGraph step")
    ap.add_argument("--surface", choices=("lightning", "driver"), default="lightning",
                    help="who drives the step.  'lightning' (default at N = 1): the reference's own surface -- Lightning 1.9's hook order "
                         "(tests/lightning_order.Trainer: training_step -> on_before_zero_grad -> optimizer_zero_grad -> backward -> "
                         "optimizer.step -> lr_scheduler_step) over train_dataloader(), a torch.optim.Adam built as train_sed.py:199-201 "
                         "builds it; SEDTask4's whole-step mode runs the captured step behind training_step (with --no-graph: the hooks "
                         "do their work one by one, eager launches).  'driver': desed_task_amd's own GraphedStepDriver / StepDriver "
                         "called directly (what N > 1 always uses: the reference's trainer refuses more than one GPU)")
    ap.add_argument("--host-batches", action="store_true",
                    help="--surface lightning only: the loader hands out PINNED HOST tensors (what a real DataLoader does), so every step "
                         "pays the 30.7 MB host-to-device copy of the next batch's waveforms + its labels: the PCIe-inclusive rate that "
                         "DESIGN.md quotes next to the headline (never the headline `value`: the metric is defined on inputs resident in HBM)")
    ap.add_argument("--embeddings", action="store_true",
                    help="secondary workload (SURVEY 8f rank 3): the 2023 'pretrained' step, frozen BEATs-shaped embeddings "
                         "(768 x 496 per clip) fused into the CRNN (confs/pretrained.yaml); not the headline metric")
    ap.add_argument("--gru-dw-atomic", action="store_true", help="A/B: BiGRU weight gradients through zero fill + atomic split-K")
    ap.add_argument("--no-gru-dw-side", action="store_true", help="A/B: BiGRU weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-cnn-dw-side", action="store_true", help="A/B: the weight-gradient GEMMs of CNN blocks 1-6 on the backward chain (= SED_CNN_DW_SIDE=0)")
    ap.add_argument("--no-park-loss", action="store_true", help="A/B: the eight loss sums inside the loss launch (fence + ticket) instead of beside the chain")
    ap.add_argument("--no-dx-splitk", action="store_true", help="A/B: the BiGRU dX products as one K slice (118 / 236 workgroups)")
    ap.add_argument("--no-defer", action="store_true", help="A/B: the head's weight-gradient sums and the BiGRU bias-gradient sums on the backward chain")
    ap.add_argument("--no-cnn-prologue", action="store_true", help="A/B: SpecAugment bands, weight packs and the copy of the hand-over features as three launches")
    ap.add_argument("--no-overlap", action="store_true",
                    help="A/B at N > 1: ONE blocking all-reduce over the whole gradient arena after backward (one graph) instead of "
                         "bucket A under the CNN backward (two graphs); same as SED_DDP_OVERLAP=0")
    ap.add_argument("--overlap", action="store_true",
                    help="A/B at N > 1: force the bucketed exchange (bucket A's asynchronous all-reduce under the CNN backward; with "
                         "--prefetch teacher that costs the one-graph structure); same as SED_DDP_OVERLAP=1")
    ap.add_argument("--gru-dw-side", action="store_true",
                    help="A/B at N > 1: force the BiGRU / CNN weight-gradient GEMMs onto the side stream as at N = 1 (the default over RCCL "
                         "with ONE all-reduce after backward; off with the bucketed overlap and over gloo, see launcher.StepDriver); same "
                         "as SED_GRU_DW_SIDE=1")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="N = 1 only: run the data-parallel step structure (graph split, bucketed RCCL all-reduces, eager Adam) on a "
                         "process group of ONE rank -- same bits as the plain step, times the exchange machinery without link time; "
                         "same as SED_DDP_REHEARSE=1")
    ap.add_argument("--prefetch", choices=("off", "tails", "backward", "teacher"), default="teacher",
                    help="software-pipelined front half: the mel kernel of batch k+1 runs on a side stream under step k's BiGRU "
                         "phases (fork before the student/teacher tails, or before backward); 'teacher': the whole front half of step "
                         "k+1 (mel, mixup, log/min-max) and the teacher's CNN forward run under step k's backward.  Every step still "
                         "computes exactly one batch's features and one teacher forward")
    ap.add_argument("--no-bn-fold", action="store_true", help="A/B: BatchNorm backward of blocks 1-6 as its own pass (sed_bn_bwd_apply)")
    ap.add_argument("--ts-probe", action="store_true",
                    help="diagnostics: GPU wall-clock stamps inside the replayed step (tools/ts_probe.py); adds ~10 one-thread kernels")
    ap.add_argument("--dump-launches", default=None, metavar="PATH",
                    help="diagnostics: write every launch shape's median time (the rows behind roofline_families) as JSON to PATH")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run the same program on the CPU emulator of the kernels over gloo at toy sizes (launch-path check "
                         "only, the numbers are meaningless)")
    ap.add_argument("--lib", default=None, help="A/B: bind another build of the C-ABI library (tools/build_variant.py)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=INT",
                    help="A/B runs: override a kernel choice of the library (desed_task_amd._lib.TUNING_KEYS), e.g. glu_bwd128_split=3")
    args = ap.parse_args()
    if args.no_overlap and args.overlap:
        raise SystemExit("--overlap and --no-overlap exclude each other")
    if args.no_overlap:
        os.environ["SED_DDP_OVERLAP"] = "0"
    if args.overlap:
        os.environ["SED_DDP_OVERLAP"] = "1"
    if os.environ.get("NCCL_DEBUG", "").upper() in ("INFO", "TRACE") and (args.gpus > 1 or args.rehearse_exchange):
        # RCCL's own account of what it runs (algorithm / protocol per collective size) goes to a file per rank that rank 0 quotes
        # in `dist.rccl_debug` -- it must be set before the communicator exists, i.e. here, before self_launch / init_distributed
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING,COLL")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/sed_bench_rccl_%p.log")
    if args.gru_dw_side:
        os.environ["SED_GRU_DW_SIDE"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU), before this process touches the GPU
        sys.exit(self_launch(args.gpus, args.dry_run))
    dry = args.dry_run
    if dry:
        global BATCH, N_SAMPLES, N_FRAMES_OUT
        BATCH, N_SAMPLES = (1, 1, 1), 8192 + 1024
        N_FRAMES_OUT = (1 + N_SAMPLES // 256) // 4
        torch.set_num_threads(1)
        sys.stderr.write("bench.py --dry-run: CPU emulator of the kernels, gloo, %d clips of %d samples -- NOT a measurement\n"
                         % (sum(BATCH), N_SAMPLES))
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X (no GPU visible); `--dry-run` checks the launch path on the CPU emulator")
    elif args.gpus > torch.cuda.device_count() and not SHARE_GPU():
        raise SystemExit("--gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))

    from desed_task_amd import _lib
    if dry:
        from tests.emu_support import bind_emulator         # test infrastructure, dry run only
        bind_emulator()
    if args.lib:
        _lib.use_library(os.path.abspath(args.lib), is_emulator=False)
    for kv in args.tuning:
        key, val = kv.split("=")
        _lib.set_tuning(key, int(val))
    if args.gru_dw_atomic:
        from desed_task_amd import ops as _ops
        _ops.GRU_DW_ATOMIC = True
    from desed_task_amd.arena import FusedAdam
    from desed_task_amd.launcher import StepDriver, init_distributed
    from desed_task_amd.nnet.CRNN import CRNN
    from desed_task_amd.sed_trainer import SEDTask4
    from desed_task_amd.utils.schedulers import ExponentialWarmup

    if args.no_bn_fold:
        from desed_task_amd import ops as _ops3
        _ops3.BN_BWD_FOLD = False
    if args.rehearse_exchange:
        if args.gpus != 1:
            raise SystemExit("--rehearse-exchange is the one-rank rehearsal of the N > 1 step: use it with --gpus 1")
        os.environ["SED_DDP_REHEARSE"] = "1"
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
    rank, local, world = init_distributed(backend="gloo" if dry else None)
    grouped = dist.is_initialized()                 # world > 1, or the one-rank rehearsal
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cpu") if dry else torch.device("cuda", local)
    torch.manual_seed(1234 + rank); np.random.seed(1234 + rank); random.seed(1234 + rank)

    config = recipe_config()
    if args.embeddings:
        from desed_task_amd.sed_trainer_pretrained import SEDTask4      # noqa: F811
        config["net"].update(use_embeddings=True, embedding_size=768, embedding_type="frame", aggregation_type="pool1d")
        config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
    student = CRNN(**config["net"]).to(dev)
    if grouped:                                     # identical initial weights on every rank
        dist.broadcast(student.arena.flat, src=0)
    lightning = args.surface == "lightning" and not grouped
    if lightning:
        opt = torch.optim.Adam(student.parameters(), 1e-3, betas=(0.9, 0.999))      # train_sed.py:199-201, verbatim; SEDTask4 adopts it
    else:
        opt = FusedAdam(student.parameters(), lr=1e-3, betas=(0.9, 0.999), arena=student.arena)
    sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 50 * 118), "interval": "step"}

    class Enc:
        labels = list(range(10))
    task = SEDTask4(config, Enc(), student, opt=opt, scheduler=sched).to(dev)
    if not isinstance(opt, FusedAdam):
        raise RuntimeError("bench: SEDTask4 did not adopt the recipe's torch.optim.Adam")
    opt.arena = task.sed_student.arena
    task.train()
    if os.environ.get("SED_OVERLAP_TAILS") is not None:
        task.overlap_tails = os.environ["SED_OVERLAP_TAILS"] == "1"
    use_graph = not args.no_graph and not dry
    driver = None
    if lightning:
        # nobody builds a driver: SEDTask4.training_step does, at the first batch the trainer loop hands it (sed_trainer.py, "whole-step
        # mode"); --no-graph: the hooks do their own work, one eager launch after the other, torch-free Adam still one launch
        task.whole_step = bool(use_graph or dry) and not args.no_graph
        task.whole_step_prefetch = args.prefetch
        task.whole_step_warmup = 3
    elif use_graph:
        # the step is captured once into a hipGraph (desed_task_amd/graph.py) and replayed: 3 eager steps, 1 capture step
        from desed_task_amd.graph import GraphedStepDriver
        driver = GraphedStepDriver(task, world_size=world, warmup=3, prefetch=args.prefetch)
    else:
        driver = StepDriver(task, world_size=world, prefetch=args.prefetch)
    pipelined = args.prefetch != "off"
    ts_probe = None
    if args.ts_probe and not dry:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from ts_probe import TsProbe
        from desed_task_amd import ops as _ops_probe
        ts_probe = TsProbe(dev)
        _ops_probe.PROBE = ts_probe
    if args.no_park_loss:
        from desed_task_amd import ops as _ops6
        _ops6.PARK_LOSS_SUMS = False
    if args.no_dx_splitk:
        from desed_task_amd import ops as _ops4
        _ops4.GRU_DX_SPLITK = False
    if args.no_defer:
        from desed_task_amd import ops as _ops3
        _ops3.DEFER_OFF_CHAIN = False
    if args.no_cnn_prologue:
        from desed_task_amd.nnet.CNN import CNN as _CNN
        _CNN.FUSE_PROLOGUE = False
    if args.no_gru_dw_side:
        from desed_task_amd import ops as _ops2
        _ops2.GRU_DW_SIDE_ALLOWED = False
    if args.no_cnn_dw_side:
        from desed_task_amd import ops as _ops7
        _ops7.CNN_DW_SIDE = False
    audio, labels = synthetic_batch(dev, 1234 + rank)
    emb = None
    if args.embeddings:
        emb = torch.randn(sum(BATCH), 768, 496, device=dev, generator=torch.Generator(device=dev).manual_seed(77 + rank))

    inputs = {"audio": audio, "emb": emb, "next_audio": audio}

    def next_labels():
        """prefetch 'teacher': the announced batch's labels.  The step only READS them (they are copied into its hand-over buffer and
        mixed there), so once the graph's static next-label buffer exists the loader's copy of the synthetic labels simply lives in
        it -- like the waveforms, resident in HBM before the timed region; before the capture: a fresh copy."""
        if args.prefetch != "teacher":
            return None
        buf = inputs.get("next_labels")
        return labels.clone() if buf is None else buf

    def one_step(i):
        # graph mode: the driver copies every batch tensor into its static input buffers, so `labels` (mixed in place by the
        # step) needs no clone; eager mode works on the tensors it is given
        captured = getattr(driver, "graph", None) is not None
        if lightning:       # (the batch tuples train_dataloader() handed out: the captured step's static buffers have their arity)
            tail = ([1.0] * sum(BATCH),) + ((inputs["emb"],) if emb is not None else ())
            batch, nxt = (inputs["audio"], labels if captured else labels.clone()) + tail, (inputs["next_audio"], next_labels()) + tail
        else:
            batch, nxt = (inputs["audio"], labels if captured else labels.clone(), None, inputs["emb"]), (inputs["next_audio"], next_labels(), None, None)
        if pipelined:       # the loader hands over batch k and announces batch k + 1 (synthetic: the same clips again)
            driver.run_step(batch, i, next_batch=nxt)
        else:
            driver.run_step(batch, i)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def adopt_static_buffers():
        # the synthetic clips already live in HBM: hand the graph's own input buffers back as the batch, like a loader that
        # writes its batches straight into them, so that no per-step staging copy of the 30 MB of audio is timed.  Round 4: the step
        # no longer mixes the announced labels in place, so they are staged once as well (round 3 re-staged 300 KB every step).
        if getattr(driver, "graph", None) is None or driver.input_buffers() is None:
            return
        bufs = driver.input_buffers()
        inputs["audio"] = bufs[0]
        if emb is not None:
            inputs["emb"] = bufs[3]
        if pipelined:       # the loader's target for the NEXT batch's waveforms; one step later the same buffer IS the batch
            inputs["audio"] = inputs["next_audio"] = driver.next_audio_buffer()
            inputs["next_labels"] = driver.next_label_buffer()
            if inputs["next_labels"] is not None:
                inputs["next_labels"].copy_(labels)

    # untimed: the W warm-up steps, plus (graph mode) whatever is still missing for the capture to lie outside the timed region
    n_untimed = max(args.warmup, 5) if use_graph else args.warmup
    graph_note = None
    surface_info = None
    out_metric_note = ""
    if lightning:
        # ---- the reference's surface: a Lightning-1.9-order loop over train_dataloader(); ONE epoch of n_untimed + K + 1 resident batches:
        # the K timed steps are steps n_untimed .. n_untimed + K - 1 of it, the epoch's last batch (no successor: the driver's eager
        # fall-back) lies behind the timed region.
        from desed_task_amd.lookahead import BatchList
        from tests.lightning_order import Trainer
        n_untimed = max(n_untimed, 8)
        pad = [1.0] * sum(BATCH)

        host = None
        if args.host_batches:
            pin = (lambda t: t.cpu().pin_memory()) if torch.cuda.is_available() else (lambda t: t.cpu())
            host = (pin(audio), pin(labels), None if emb is None else pin(emb))

        class Resident(BatchList):
            def __getitem__(self, i):
                if host is not None:        # a fresh host batch every time, like a DataLoader's collate output in pinned memory
                    return (host[0], host[1], pad) + ((host[2],) if emb is not None else ())
                lab = inputs.get("next_labels")
                return (inputs["audio"], labels.clone() if lab is None else lab, pad) + ((inputs["emb"],) if emb is not None else ())

        task.train_data = Resident([None] * (n_untimed + args.steps + 1))
        marks = {}

        def on_step(tr, model, i):
            nonlocal driver
            n = i + 1
            driver = task._driver
            if n == n_untimed - 3 and host is None:
                adopt_static_buffers()              # (batches the look-ahead has already fetched still carry the old tensors: 2 steps)
            if n == n_untimed:
                sync()
                if not task.sed_student.arena.grads_are_flat():
                    raise RuntimeError("bench: a parameter's .grad is not the arena's view after the warm-up steps")
                sync()
                if getattr(driver, "graph", None) is not None:
                    marks["fb0"], marks["rp0"] = driver.eager_fallbacks, driver.reprimes
                marks["t0"] = time.perf_counter()
            elif n == n_untimed + args.steps:
                sync()
                marks["t1"] = time.perf_counter()
                if driver is not None and getattr(driver, "graph", None) is not None:
                    marks["fallbacks"], marks["reprimes"] = driver.eager_fallbacks, driver.reprimes

        Trainer(max_epochs=1, on_step=on_step).fit(task)
        sync()
        dt = dt_local = marks["t1"] - marks["t0"]
        driver = task._driver
        use_graph = driver is not None and getattr(driver, "graph", None) is not None
        if task.whole_step and not dry and not use_graph:
            raise RuntimeError("bench: the whole-step mode never captured its graph")
        surface_info = {"surface": "lightning", "host_batches": bool(args.host_batches),
                        "loop": "tests/lightning_order.Trainer: Lightning 1.9's automatic-optimisation hook order over train_dataloader() "
                                "(one epoch of %d resident batches), optimizer built as torch.optim.Adam(student.parameters(), 1e-3, "
                                "betas=(0.9, 0.999))" % (n_untimed + args.steps + 1),
                        "whole_step": bool(task.whole_step),
                        "eager_fallbacks_in_timed_region": marks["fallbacks"] - marks["fb0"] if "fb0" in marks else None,
                        "reprimes_in_timed_region": marks["reprimes"] - marks["rp0"] if "rp0" in marks else None}
        if host is not None:
            out_metric_note = "PCIe-INCLUSIVE (pinned host batches, one host-to-device copy of the next batch per step) "
        if driver is None:
            # hooks mode (--no-graph): the per-launch timing below still needs a StepDriver for its eager steps
            driver = StepDriver(task, world_size=1, prefetch="off")
            pipelined = False
        else:
            # the same captured step driven by hand (StepDriver protocol, what --surface driver times): K more steps, after the epoch's
            # last batch has been run eagerly -- 3 untimed steps re-prime the pipeline
            for i in range(3):
                one_step(i)
            sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                one_step(i)
            sync()
            surface_info["driver_surface_ms_per_step"] = round((time.perf_counter() - t0) / args.steps * 1e3, 3)
            surface_info["lightning_over_driver"] = round(dt / args.steps * 1e3 / surface_info["driver_surface_ms_per_step"], 4)
    else:
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
        if use_graph:
            adopt_static_buffers()
        sync()
        # the optimizer's one-launch path needs every gradient in the parameter arena (a gradient autograd had to clone would send Adam
        # down the per-tensor path and, with side-stream producers, read stale values): checked on the warm-up steps' result, before the clock
        if not task.sed_student.arena.grads_are_flat():
            raise RuntimeError("bench: a parameter's .grad is not the arena's view after the warm-up steps")
        if grouped:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        sync()
        dt_local = time.perf_counter() - t0             # this rank's own K steps (before it waits for the others)
    ts_rows = None
    if ts_probe is not None:
        ts_rows = ts_probe.read()
        from desed_task_amd import ops as _ops_probe
        _ops_probe.PROBE = None
    if not lightning:
        if grouped:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
    # Data-parallel runs: PROBE_STEPS more steps of the SAME launch path (graph replays + eager exchange tail, or eager steps) with
    # marks around the exchange tail -- after the timed region, so the K timed steps carry no extra events.
    exchange_probe = None
    if grouped:
        from desed_task_amd.launcher import ExchangeProbe
        PROBE_STEPS = 2 if dry else 7
        d0 = driver.eager if use_graph else driver
        d0.probe = exchange_probe = ExchangeProbe(dev)
        for i in range(PROBE_STEPS):
            one_step(args.warmup + args.steps + i)
        sync()
        d0.probe = None
        exchange_probe.steps = exchange_probe.steps[1:]         # (the first one absorbs the barrier's wake-up)
    # Per-launch HIP events cannot be placed inside a graph replay, and bracketing all ~330 launches of a step with events
    # would perturb the timed region: the kernels are timed over EAGER_STEPS eager steps of the same workload right after the
    # timed region (same process, same tensors, every launch bracketed by events on the stream it is launched on; the first of
    # them is not recorded -- it re-warms the eager path after the graph replays -- and every entry reports its MEDIAN launch).
    EAGER_STEPS = 5
    timer = KernelTimer(None)
    eager = driver.eager if use_graph else driver
    if not dry:
        def eager_step(i):
            batch = (inputs["audio"], labels.clone(), None, emb)
            if pipelined:
                eager.run_step(batch, i, next_batch=(inputs["next_audio"], next_labels(), None, None))
            else:
                eager.run_step(batch, i)

        eager_step(0)
        torch.cuda.synchronize()
        timer.wrap(_lib.get())
        for i in range(EAGER_STEPS):
            eager_step(i)
        torch.cuda.synchronize()
        timer.unwrap()
    dist_info = None
    if grouped:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_local / args.steps * 1e3], device=dev, dtype=torch.float64)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank = [round(float(v.item()), 3) for v in per_rank]
        d0 = driver.eager if use_graph else driver
        scheme = "eager launches"
        if use_graph:
            scheme = ("two graphs [forwards + losses + EMA + backward of heads/BiGRU | backward of the CNN], bucket A's all-reduce "
                      "issued between the replays" if getattr(driver, "graph_cnn", None) is not None
                      else "one graph up to the end of backward")
            scheme += "; all-reduce(s) and Adam eager"
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_per_device": "1 (rank r -> device r)",
                     "overlap_allreduce": bool(d0.overlap), "gru_dw_side_stream": bool(d0.gru_dw_side), "graph_scheme": scheme,
                     "bucket_log": [[tag, lo, round(4 * n / 1e6, 3)] for tag, lo, n in d0.bucket_log],
                     "bucket_log_fields": "[bucket, first float of the gradient arena, MB] of the last step's collectives, in issue order",
                     "ms_per_step_per_rank": per_rank, "ms_per_step_rank_min": min(per_rank), "ms_per_step_rank_max": max(per_rank),
                     # measured on rank 0 over the probe steps that follow the timed region (launcher.ExchangeProbe): medians, us
                     "exchange_tail_us": exchange_probe.summary() if exchange_probe is not None else None,
                     "exchange_tail_fields": "segment -> {device_us: HIP-event time on the compute stream (null on the CPU emulator), "
                                             "host_us: host clock between the same two points}.  'exposed_exchange' = end of backward "
                                             "(of the replayed graph) -> gradients reduced = what the collective(s) add to a step; "
                                             "host_us > device_us there means the host, not the link, sets the gap before Adam",
                     "expected": "N ranks ~ N x (1-GPU clips/s) x t1 / (t1 + exposed_exchange): the step has no other cross-rank work; "
                                 "falsified by exposed_exchange.device_us >> 100 (a 4.45 MB all-reduce should be latency-bound) or by "
                                 "ms_per_step_rank_max - ms_per_step_rank_min > 5 % (a straggling rank, not the exchange)",
                     "rccl_debug": rccl_debug_lines()}
    loss_val = float(task.logged["train/student/loss_strong"])
    backend_name = dist.get_backend() if grouped else None
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    step_ms = dt / args.steps * 1e3
    precision = task.sed_student.cnn.conv_precision
    rows, families = roofline_tables(timer.summary(), EAGER_STEPS, step_ms, precision)
    if args.dump_launches:
        with open(args.dump_launches, "w") as fh:
            json.dump([{k: r[k] for k in ("entry", "shape", "launches_per_step", "avg_us", "us_per_step")} for r in rows], fh)
    roofline = None
    if rows:
        # the dominant kernel = the launch shape with the largest total time per step in the whole-step event trace
        dom = rows[0]
        kname = kernel_name(dom)
        issue = 3.0 if (dom["bound"] == "mfma" and dom["peak"] == PEAK_BF16_MFMA_TFLOPS) else 1.0
        roofline = {"bound": dom["bound"], "kernel": kname, "entry": dom["entry"], "shape": dom["shape"],
                    "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                    "traffic": pmc_traffic(kname),
                    "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_us"],
                    "share_of_step": round(dom["us_per_step"] / (step_ms * 1e3), 4),
                    "algorithmic_work_per_launch": dom["work"],
                    "note": "dominant = largest total time per step over ALL entry points (HIP events around every launch, %d eager steps "
                            "right after the timed region); achieved = algorithmic %s per launch / median launch time.%s%s"
                            % (EAGER_STEPS, "bytes" if dom["bound"] == "hbm" else "FLOPs",
                               " Split-bf16 kernel: the MFMA pipe issues 3 bf16 MFMAs per algorithmic product (mfma_issue_frac)."
                               if issue == 3.0 else "",
                               " The GRU recurrence runs on the f32 vector pipes (packed FMA), whose peak equals the exact-f32 MFMA peak "
                               "of 157.3 TFLOP/s; it is bound by the latency of 156 dependent steps, not by a throughput roof."
                               if dom["bound"] == "valu" else ""),
                    "traffic_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
                                    "(profiles/*_pmc_traffic.json, gfx950 FETCH_SIZE half-count correction), or null"}
        if issue == 3.0:
            roofline["mfma_issue_frac"] = round(3.0 * dom["frac"], 4)
        # kept from round 1: the heaviest 3x3 convolution launch as its own entry
        convs = [r for r in rows if r["family"] == "conv"]
        if convs:
            c = max(convs, key=lambda r: r["work"])
            roofline["conv_entry"] = {k: c[k] for k in ("entry", "shape", "avg_us", "achieved", "peak", "unit", "frac")}
            roofline["conv_entry"]["traffic"] = pmc_traffic(kernel_name(c))
    clips = sum(BATCH) * world * args.steps
    families["_note"] = ("us_per_step = sum over the family's launch shapes of launches x MEDIAN launch time, from HIP events around every "
                         "launch of %d EAGER steps after the timed region; share_of_step divides by the REPLAYED step time, and the "
                         "student / teacher tails, the EMA and the BiGRU weight-gradient GEMMs overlap on side streams, so the shares "
                         "add up to more than 1" % EAGER_STEPS)
    hbm_step, hbm_src = pmc_step_traffic()
    out = {
        "metric": ("DRY RUN (CPU emulator, toy sizes, NOT a measurement) " if dry else "") + out_metric_note + "10s-clips/sec CRNN mean-teacher train @batch48",
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
                             else (graph_note or "eager launches"), "untimed_steps": n_untimed,
                   "surface": surface_info if surface_info is not None else {
                       "surface": "driver", "loop": "desed_task_amd.graph.GraphedStepDriver / launcher.StepDriver.run_step(batch, i, "
                                                    "next_batch) called directly" + (" (N > 1 or the one-rank rehearsal: the reference's own "
                                                    "trainer refuses more than one GPU, train_sed.py:269-276)" if grouped else "")},
                   # the synthetic labels of the announced batch live in the graph's static next-label buffer (staged once,
                   # like the waveforms); a real loader pays one ~300 KB host-to-device copy per step on top of the 30.7 MB of audio
                   "labels_resident": bool(use_graph and args.prefetch == "teacher"),
                   "backend": backend_name, "world_size": world,
                   "front_end": ("mel of batch k at the head of step k" if not pipelined else
                                 "pipelined: front half of step k+1 (mel, mixup, log/min-max) + the teacher's CNN forward on a side stream "
                                 "under step k's backward; one batch's features and one teacher forward per step" if args.prefetch == "teacher"
                                 else "pipelined: mel of batch k+1 on a side stream under step k (fork before %s); one batch's features "
                                      "per step" % args.prefetch)},
        "roofline": roofline,
        # every kernel family of the step: time per step (events, eager launches), share of the replayed step, and achieved
        # / peak of its algorithmic work against the roof that bounds it
        "roofline_families": families,
        "roofline_top_launches": [{k: r[k] for k in ("entry", "shape", "bound", "launches_per_step", "avg_us", "us_per_step", "achieved",
                                                       "peak", "unit", "frac")} for r in rows[:12]],
        # SURVEY 8(d) step-level yardsticks (algorithmic work per clip x measured clips/s, per GPU)
        "step_roofline": {
            "mfma_tflops": round(6.464e9 * clips / dt / world / 1e12, 2),
            "mfma_frac_of_f32_peak": round(6.464e9 * clips / dt / world / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            "mfma_frac_of_bf16_peak": round(6.464e9 * clips / dt / world / (PEAK_BF16_MFMA_TFLOPS * 1e12), 5),
            "mel_hbm_frac": round(960512.0 * clips / dt / world / 8.0e12, 6),
            "hbm_bytes_per_step": hbm_step, "hbm_bytes_per_step_source": hbm_src, "lib_rev": lib_rev(),
            "hbm_note": None if hbm_step else "no profiles/*_pmc_traffic.json was taken on this build of the library (lib_rev): null, not a stale figure",
            "hbm_frac_of_8TBs": round(hbm_step / (step_ms * 1e-3) / 8.0e12, 4) if hbm_step else None,
            "note": "6.464 GFLOP/clip = conv1-6 + GLU1-6, student fwd + dgrad + wgrad + teacher fwd; 960 512 B/clip = mel path in+out; "
                    "whole-step clips/s, so both are diluted by the GRU recurrence and the HBM-bound narrow blocks (DESIGN.md 8)"},
    }
    if ts_rows is not None:
        out["ts_probe_us"] = [[t, round(u, 1)] for t, u in ts_rows]
        sys.stderr.write("in-graph timestamps of the last replayed step (us since step_start):\n" + "".join("  %-16s %9.1f\n" % (t, u) for t, u in ts_rows))
    if dist_info is not None:
        out["dist"] = dist_info
        if args.rehearse_exchange:
            out["dist"]["rehearsal"] = ("ONE rank: the N > 1 step structure (graph split, all-reduces over a one-rank communicator, eager "
                                        "Adam) on this GPU; bit-identical to the plain step, no link time -- NOT a scaling measurement")
    if world > 1 and torch.cuda.is_available() and world > torch.cuda.device_count():
        out["metric"] = "SHARED GPU (%d gloo ranks on %d device(s): launch-path check, NOT a measurement) " % (world, torch.cuda.device_count()) + out["metric"]
        out["shared_gpu"] = True
        if dist_info is not None:
            out["dist"]["ranks_per_device"] = "%d ranks time-slice %d device(s) over gloo" % (world, torch.cuda.device_count())
    if dry:
        out["dry_run"] = True
        out["data"] = "synthetic, toy sizes (%d clips of %d samples per rank) on the CPU emulator" % (sum(BATCH), N_SAMPLES)
    if world == 1 and not args.no_cpu_baseline and not args.embeddings and not dry:
        out["cpu_baseline"] = cpu_baseline_bounded()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # whatever native libraries left in C stdio buffers goes out BEFORE the JSON line
    except OSError:
        pass
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
