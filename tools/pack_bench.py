"""Times the CNN prologue launch (weight packs + SpecAugment bands + copy of the hand-over features) at the benchmarked shapes
(diagnostics): python tools/pack_bench.py [lib=<other build>]"""
import os
import sys
import torch
sys.path.insert(0, ".")
from desed_task_amd import _lib, ops, features
for a in sys.argv[1:]:
    if a.startswith("lib="):
        _lib.use_library(os.path.abspath(a[4:]), is_emulator=False)
dev = "cuda"
shapes = [(32, 16), (64, 32), (128, 64), (128, 128), (128, 128), (128, 128)]
ws = [torch.randn(co, ci, 3, 3, device=dev) * 0.1 for co, ci in shapes]
x = torch.randn(48, 626, 128, device=dev)


def run(student):
    pro = {"bounds": dict(features.specaug_request(48, 128, 626, 10, 0.2, 5, 0.2, True, 1234), out=torch.empty(48, 4, dtype=torch.int32, device=dev))}
    if student:
        pro["copy"] = (x, torch.empty_like(x))
    return ops.pack_conv_weights(ws, student, "bf16x3", prologue=pro)


for student in (True, False):
    ts = []
    for it in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(student); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("student" if student else "teacher", "prologue: median %.1f us, min %.1f us" % (sorted(ts)[10], min(ts)))
