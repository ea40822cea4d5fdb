#!/usr/bin/env python3
"""Counter-check of tests/test_gpu_ddp_graph.py::test_capture_next_to_live_rccl_collectives: the SAME worker with graph.quiesce_collectives
reduced to round 3's behaviour (device synchronised, the RCCL watchdog not drained).  Each run = 30 captures of the training step right
behind live collectives; a run that dies shows torch.multiprocessing's `EOFError: Ran out of input` (the watchdog's event query during the
capture aborted the worker) -- round 3's "one run in 43".  Measured on one MI355X box: 1 of 4 runs dead without the drain, 0 with it.
    python tools/nodrain_check.py [RUNS=4]          (needs the GPU)"""
import os, sys, tempfile
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_ddp_graph as T
import desed_task_amd.graph as G


def worker(rank, port, out, reps):
    G.quiesce_collectives = lambda dev: torch.cuda.synchronize(dev)      # round 3's behaviour: device drained, watchdog not
    T._capture_worker(rank, port, out, reps)


if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dead = 0
    for i in range(runs):
        out = tempfile.mkdtemp()
        try:
            mp.spawn(worker, args=(T._free_port(), out, 30), nprocs=1, join=True)
            print("run", i, "survived", torch.load(os.path.join(out, "capture.pt")))
        except Exception as e:  # noqa: BLE001
            dead += 1
            print("run", i, "DIED:", type(e).__name__, str(e).splitlines()[0][:120])
    print("dead %d of %d" % (dead, runs))
