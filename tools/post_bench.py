"""Timing of the inference post-processing at the recipe's test-time shape: 48 clips x 156 frames x 10 classes,
median window 7, the 50 PSDS thresholds + 0.5.  Device path = desed_task_amd.postprocess (two launches + one copy per batch);
CPU path = the oracle restatement of the reference's per-clip loop (scipy median filter, one threshold at a time,
find_contiguous_regions per class).  Usage (GPU box): python tools/post_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from desed_task_amd import postprocess as PP  # noqa: E402
from oracle import sed_oracle as O  # noqa: E402
from tests.parity_cases import _Encoder  # noqa: E402

B, T, NC = 48, 156, 10
thresholds = list(np.arange(1 / 100, 1, 1 / 50)) + [0.5]
enc = _Encoder(["c%d" % i for i in range(NC)], audio_len=10.0)
g = torch.Generator().manual_seed(0)
# smooth posteriors (random walks through a sigmoid): a realistic number of events per clip
walk = torch.cumsum(torch.randn(B, NC, T, generator=g) * 0.6, -1)
strong = torch.sigmoid(walk - walk.mean(-1, keepdim=True)).cuda()
files = ["/d/t%d.wav" % i for i in range(B)]

for _ in range(3):
    PP.batched_decode_preds(strong, files, enc, median_filter=7, thresholds=thresholds)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    _, _, dfs = PP.batched_decode_preds(strong, files, enc, median_filter=7, thresholds=thresholds)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / n * 1e3
# kernels only
scores = strong.transpose(1, 2).contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    filt = PP.median_filter_scores(scores, 7)
    PP.threshold_events(filt, thresholds)
e1.record(); torch.cuda.synchronize()
ker_ms = e0.elapsed_time(e1) / n

s_np = strong.cpu().numpy()
t0 = time.perf_counter()
n_ev = 0
for j in range(B):
    filt = O.median_filter_scores(s_np[j].T, 7)
    for th in thresholds:
        n_ev += len(O.decode_events(filt, np.float32(th)))
cpu_ms = (time.perf_counter() - t0) * 1e3
n_dev = sum(len(d) for d in dfs.values())
assert n_dev == n_ev, (n_dev, n_ev)
print("batch of %d clips, %d thresholds, %d events: device path %.2f ms end to end (median + threshold/region kernels + copy %.3f ms, "
      "rest = DataFrame assembly); reference-style per-clip loop on the host %.1f ms -> %.0fx" % (B, len(thresholds), n_ev, dev_ms, ker_ms, cpu_ms, cpu_ms / dev_ms))
