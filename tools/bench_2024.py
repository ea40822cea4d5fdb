"""Throughput of the 2024 recipe's training step (BASELINE config 4's CRNN half) at the recipe's own sizes:
`recipes/dcase2024_task4_baseline/confs/pretrained.yaml` -- batch [12, 6, 6, 12, 24] = 60 clips of 10 s, 27 classes, n_RNN_cell 192,
dropout 0.5, dropstep_recurrent 0.3 x 16, frozen 768 x 496 embeddings per clip, mixup on features and embeddings, class masks --
random weights, synthetic data.  Secondary workload: not the headline metric of bench.py.   python tools/bench_2024.py [--graph]"""
import json, os, random, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from desed_task_amd.arena import FusedAdam
from desed_task_amd.launcher import StepDriver
from desed_task_amd.nnet.CRNN import CRNN
from desed_task_amd.sed_trainer_pretrained_2024 import SEDTask4
from desed_task_amd.utils.schedulers import ExponentialWarmup
from bench import recipe_config

BS, NCLASS = (12, 6, 6, 12, 24), 27
dev = torch.device("cuda", 0)
torch.manual_seed(1); np.random.seed(1); random.seed(1)
config = recipe_config()
config["training"].update(batch_size=list(BS), mixup="soft", mixup_prob=0.5, epoch_decay=100, const_max=2)
config["net"].update(dropout=0.5, rnn_layers=1, nclass=NCLASS, n_RNN_cell=192, dropstep_recurrent=0.3, dropstep_recurrent_len=16,
                     use_embeddings=True, embedding_size=768, embedding_type="frame", aggregation_type="pool1d")
config["pretrained"] = {"e2e": False, "freezed": True, "model": "beats"}
student = CRNN(**config["net"]).to(dev)
opt = FusedAdam(student.parameters(), lr=1e-3, betas=(0.9, 0.999), arena=student.arena)
sched = {"scheduler": ExponentialWarmup(opt, 1e-3, 50 * 118), "interval": "step"}


class Enc:
    labels = list(range(NCLASS))


task = SEDTask4(config, Enc(), student, None, opt=opt, scheduler=sched).to(dev)
opt.arena = task.sed_student.arena
task.train()
B = sum(BS)
g = torch.Generator(device=dev).manual_seed(5)
audio = 0.1 * torch.randn(B, 160000, device=dev, generator=g)
labels = (torch.rand(B, NCLASS, 156, device=dev, generator=g) < 0.1).float()
ns = BS[0] + BS[1] + BS[2]
labels[ns:ns + BS[3], :, 1:] = 0.0
labels[ns + BS[3]:] = 0.0
emb = torch.randn(B, 768, 496, device=dev, generator=g)
valid = torch.zeros(B, NCLASS, dtype=torch.bool, device=dev)
valid[:BS[0], 10:] = True
valid[BS[0]:, :10] = True
# --prefetch: the pipelined front half (round 4): mel, per-data-set mixup of features and embeddings, log / min-max and the teacher's CNN
# forward of step k + 1 under step k's backward; the announced batch is only read, so the resident synthetic tensors are passed as they are
pipelined = "--prefetch" in sys.argv
pf = "teacher" if pipelined else None
if "--graph" in sys.argv:
    from desed_task_amd.graph import GraphedStepDriver
    driver = GraphedStepDriver(task, world_size=1, warmup=3, prefetch=pf)
else:
    driver = StepDriver(task, world_size=1, prefetch=pf)
W, K = 8, 20


def step(i):
    if pipelined:
        return driver.run_step((audio, labels, None, emb, valid), i, next_batch=(audio, labels, None, emb, valid))
    return driver.run_step((audio, labels.clone(), None, emb, valid), i)


for i in range(W):
    step(i)
if pipelined and "--graph" in sys.argv:
    # like a loader that writes its batches straight into the graph's input buffers: no per-step staging copies of resident tensors
    audio = driver.next_audio_buffer()
    labels = driver.next_label_buffer()
    emb = driver.next_extra_buffers()["embeddings"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    loss = step(W + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
# per-entry roofline rows like bench.py's: HIP events around every launch of 3 eager steps after the timed region
# (the 2024 step's entries are the CRNN step's + sed_embcat: bench.py's work table applies; shapes with n_RNN_cell = 192)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import KernelTimer, roofline_tables, PEAK_BF16_MFMA_TFLOPS
from desed_task_amd import _lib as _lib_
eager = driver.eager if hasattr(driver, "eager") else driver
timer = KernelTimer(None)
EAGER = 3


def eager_step(i):
    if pipelined:
        return eager.run_step((audio, labels, None, emb, valid), i, next_batch=(audio, labels, None, emb, valid))
    return eager.run_step((audio, labels.clone(), None, emb, valid), i)


eager_step(0); torch.cuda.synchronize()
timer.wrap(_lib_.get())
for i in range(EAGER):
    eager_step(i)
torch.cuda.synchronize()
timer.unwrap()
rows, families = roofline_tables(timer.summary(), EAGER, dt * 1e3, task.sed_student.cnn.conv_precision)
clips_s = B / dt
print(json.dumps({"workload": "dcase2024 pretrained.yaml training step: batch 60 = [12,6,6,12,24] x 10 s, 27 classes, n_RNN_cell 192, "
                              "768 x 496 embeddings per clip, dropout + dropstep + mixup on", "launch": "hipGraph" if "--graph" in sys.argv else "eager",
                  "front_end": "pipelined (teacher)" if pipelined else "inline",
                  "ms_per_step": round(dt * 1e3, 3), "clips_per_s": round(B / dt, 1), "loss": round(float(loss), 5),
                  "step_roofline": {"mfma_tflops": round(6.464e9 * clips_s / 1e12, 2),
                                    "mfma_frac_of_bf16_peak": round(6.464e9 * clips_s / (PEAK_BF16_MFMA_TFLOPS * 1e12), 5),
                                    "note": "6.464 GFLOP/clip = conv1-6 + GLU1-6 (student fwd + dgrad + wgrad + teacher fwd), the same CNN as the 2023 recipe"},
                  "roofline_families": families,
                  "roofline_top_launches": [{k: r[k] for k in ("entry", "shape", "bound", "launches_per_step", "avg_us", "us_per_step", "achieved",
                                                                 "peak", "unit", "frac")} for r in rows[:8]]}))
