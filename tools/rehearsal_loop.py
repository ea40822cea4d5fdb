#!/usr/bin/env python3
"""Root-cause tool for VERDICT r03 weak #1: the one-rank RCCL rehearsal of the data-parallel step must equal the plain step bit for
bit; one run of 43 in round 3 did not end green.  This loops the test's own body

    plain GraphedStepDriver (one graph incl. Adam)   vs   rehearsal (graph up to the end of backward + RCCL all-reduce + eager Adam)

`REPS` times in each of `PROCS` fresh processes (a fresh process = fresh RCCL communicator, fresh allocator, fresh graph pool) and
reports, per repetition, whether the final student / teacher arenas are bit-identical -- and, in `--stages` mode, the first step and
buffer whose checksum differs (gradient arena after the graph / after the all-reduce, every hand-over buffer of the pipelined front
half, both weight arenas after Adam).  A child that dies without a Python exception (signal, abort) is reported with its exit code
and the tail of its stderr (faulthandler is on): the round-3 failure was such a death, not a numeric difference (the log shows
torch.multiprocessing's EOFError on an empty error file).

    python tools/rehearsal_loop.py [--procs 8] [--reps 25] [--prefetch teacher|none] [--overlap 0|1|default] [--stages]
                                   [--env K=V ...]          (needs the GPU: tools/gpu.sh 900 'python tools/rehearsal_loop.py ...')
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _h(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:12]


def child(args):
    import faulthandler
    faulthandler.enable(all_threads=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      SED_DDP_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("SED_DIST_BACKEND", None)
    if args.overlap == "default":
        os.environ.pop("SED_DDP_OVERLAP", None)
    else:
        os.environ["SED_DDP_OVERLAP"] = args.overlap
    import random
    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import sed_oracle as O          # (diagnostics tool: synthetic inputs only)
    from tests import parity_cases as P
    from desed_task_amd import ops as _ops
    from desed_task_amd.graph import GraphedStepDriver
    from desed_task_amd.launcher import init_distributed
    init_distributed()
    assert dist.is_initialized() and dist.get_backend() == "nccl"
    dev = "cuda"
    prefetch = None if args.prefetch == "none" else args.prefetch
    bs, n_samp, steps = (1, 1, 2), 16000 + 1024, args.steps
    sd = O.make_state_dict(seed=7)
    audio = [P.to(dev, O.synth_audio(4, n_samp, seed=100 + k)) for k in range(steps)]
    labels = [P.to(dev, O.synth_labels(bs, 10, (1 + n_samp // 256) // 4, seed=5 + k)) for k in range(steps)]
    first_plain = None
    for rep in range(args.reps):
        finals, stages = [], []
        for mode in ("plain", "rehearsal"):
            os.environ["SED_DDP_REHEARSE"] = "1" if mode == "rehearsal" else "0"
            task = P.build_task(dev, bs, sd, dropout=0.5, specaug=True, rampup=5)
            driver = GraphedStepDriver(task, 1, warmup=1, prefetch=prefetch)
            assert driver.eager.exchange == (mode == "rehearsal")
            random.seed(40); np.random.seed(100); torch.manual_seed(100); torch.cuda.manual_seed(100)
            _ops.reseed_dropout()
            st = []
            for step in range(steps):
                batch = (audio[step], labels[step].clone(), None, None)
                if prefetch is None:
                    driver.run_step(batch, step)
                else:
                    nxt = (audio[step + 1], labels[step + 1].clone(), None, None) if step + 1 < steps else None
                    driver.run_step(batch, step, next_batch=nxt)
                if args.stages:
                    torch.cuda.synchronize()
                    rec = {"grad": _h(task.sed_student.arena.flat_grad), "student": _h(task.sed_student.arena.flat),
                           "teacher": _h(task.sed_teacher.arena.flat)}
                    pro = getattr(task, "_pro", None)
                    if pro is not None:
                        for k in ("x", "ht", "labels", "labels_weak"):
                            rec["pro_" + k] = _h(pro[k])
                    for i in range(7):
                        for who, model in (("s", task.sed_student), ("t", task.sed_teacher)):
                            bn = getattr(model.cnn.cnn, "batchnorm%d" % i)
                            rec["bn%d%s" % (i, who)] = _h(torch.cat([bn.running_mean, bn.running_var]))
                    st.append(rec)
            torch.cuda.synchronize()
            finals.append(torch.cat([task.sed_student.arena.flat.detach().cpu(), task.sed_teacher.arena.flat.detach().cpu()]))
            stages.append(st)
            del driver, task
        same = torch.equal(finals[0], finals[1])
        if first_plain is None:
            first_plain = finals[0]
        res = {"rep": rep, "equal": bool(same), "plain_stable": bool(torch.equal(first_plain, finals[0]))}
        if not same:
            d = (finals[0] - finals[1]).abs()
            res["max"] = d.max().item()
            res["n_diff"] = int((d > 0).sum())
        if args.stages:
            for s, (a, b) in enumerate(zip(*stages)):
                bad = [k for k in a if a[k] != b[k]]
                if bad:
                    res["first_diff"] = {"step": s, "buffers": bad}
                    break
        print("REP " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("CHILD_DONE", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=25)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--prefetch", default="teacher")
    ap.add_argument("--overlap", default="default")
    ap.add_argument("--stages", action="store_true")
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--port", type=int, default=29611)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    env = dict(os.environ)
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    total = {"procs": 0, "reps": 0, "unequal": 0, "plain_unstable": 0, "dead": [], "first_diffs": []}
    t0 = time.time()
    for p in range(args.procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--reps", str(args.reps), "--steps", str(args.steps),
               "--prefetch", args.prefetch, "--overlap", args.overlap, "--port", str(args.port + p)] + (["--stages"] if args.stages else [])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        total["procs"] += 1
        done = "CHILD_DONE" in r.stdout
        for line in r.stdout.splitlines():
            if line.startswith("REP "):
                d = json.loads(line[4:])
                total["reps"] += 1
                total["unequal"] += 0 if d["equal"] else 1
                total["plain_unstable"] += 0 if d["plain_stable"] else 1
                if not d["equal"]:
                    total["first_diffs"].append({"proc": p, **d})
        if r.returncode != 0 or not done:
            total["dead"].append({"proc": p, "returncode": r.returncode, "reps_done": sum(1 for l in r.stdout.splitlines() if l.startswith("REP ")),
                                  "stderr_tail": r.stderr[-3000:]})
        print("[proc %d] rc %d, %d reps so far, %d unequal, %d dead (%.0f s)" % (p, r.returncode, total["reps"], total["unequal"],
                                                                              len(total["dead"]), time.time() - t0), flush=True)
    total["config"] = {"prefetch": args.prefetch, "overlap": args.overlap, "stages": args.stages, "env": args.env, "steps": args.steps}
    total["seconds"] = round(time.time() - t0, 1)
    print("SUMMARY " + json.dumps(total), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(total, f, indent=1)


if __name__ == "__main__":
    main()
