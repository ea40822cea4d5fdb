"""TEST INFRASTRUCTURE ONLY: compile desed_task_amd/csrc/*.hip as host C++ against the fiber
emulator (tests/emu/hip_emu.h) -> tests/emu/libsed_emu.so.  Used by the CPU tests to check kernel
logic without a GPU; never loaded by the product path."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "desed_task_amd", "csrc")
LIB = os.path.join(HERE, "libsed_emu.so")
OBJ = os.path.join(HERE, "_obj")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/bin/amdclang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-pthread", "-DSED_EMU", "-I", HERE, "-I", CSRC, "-ffp-contract=off",
         "-Wno-unused-function", "-Wno-unknown-attributes"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "hip_emu.h")]
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")) + [os.path.join(HERE, "hip_emu.cpp")]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([CXX] + FLAGS + ["-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu compile failed for %s:\n%s" % (s, r.stderr[-6000:]))

    if jobs:
        if verbose:
            print("[emu] compiling", [os.path.basename(s) for s, _ in jobs], flush=True)
        with ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(cc, jobs))
    if jobs or _stale(LIB, objs):
        r = subprocess.run([CXX, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emu link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
