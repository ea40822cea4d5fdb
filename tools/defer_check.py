"""Diagnostics: the step with the small reductions deferred onto the side stream (ops.DEFER_OFF_CHAIN) vs on the chain -- same seeds,
eager and hipGraph; prints the first differing parameters."""
import random
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import parity_cases as P
from desed_task_amd import ops, graph as G
from desed_task_amd.launcher import StepDriver
O = P.O
dev = "cuda"
bs = (2, 2, 4)
n_samp = 16000 + 1024


def run(defer, graph):
    ops.DEFER_OFF_CHAIN = defer
    task = P.build_task(dev, bs, O.make_state_dict(seed=7), dropout=0.5, specaug=True, rampup=5)
    d = G.GraphedStepDriver(task, world_size=1, warmup=1) if graph else StepDriver(task, world_size=1)
    audio = P.to(dev, O.synth_audio(sum(bs), n_samp, seed=100))
    labels = P.to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5))
    random.seed(40); np.random.seed(100); torch.manual_seed(100); torch.cuda.manual_seed(100)
    ops.reseed_dropout()
    for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
        d.run_step((audio, labels.clone(), None, None), step)
    torch.cuda.synchronize()
    a = task.sed_student.arena
    return a.flat.detach().cpu().clone(), a.flat_grad.detach().cpu().clone(), [(n, p.numel()) for n, p in task.sed_student.named_parameters()]


for graph in (False, True):
    w0, g0, names = run(False, graph)
    w1, g1, _ = run(True, graph)
    print("graph" if graph else "eager", "weights max diff %.3e, grads max diff %.3e" % ((w0 - w1).abs().max().item(), (g0 - g1).abs().max().item()))
    off = 0
    for n, k in names:
        dg = (g0[off:off + k] - g1[off:off + k]).abs().max().item()
        dw = (w0[off:off + k] - w1[off:off + k]).abs()
        if dg > 0 or dw.max().item() > 0:
            print("   ", n, "grad diff %.3e of max %.3e; weights: %d of %d differ, max %.3e" % (dg, g0[off:off + k].abs().max().item(), int((dw > 0).sum()), k, dw.max().item()))
        off += k
