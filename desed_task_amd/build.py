"""Build recipe for the gfx950 HIP extension (C-ABI shared library, in-tree).

    python -m desed_task_amd.build          # -> desed_task_amd/libsed_hip.so

hipcc cross-compiles for gfx950 without a GPU.  One object per .hip source, compiled in parallel and
cached by mtime, then linked into ONE shared library.  Everything is compiled with -fvisibility=hidden; the
`SED_API` (extern "C", default visibility) entry points declared in include/sed_hip.h are the only exported
symbols: a linker version script (`global: sed_*; local: *`) also hides the kernel host stubs hipcc insists on
exporting (tests/test_abi.py checks the dynamic symbol table against the header).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsed_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I", CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def source_rev():
    """12 hex digits over the kernel sources (every .hip / .h under csrc/, names and contents): the identity of a build of the library.
    PMC summaries under profiles/ carry it (tools/pmc_traffic_json.py), and bench.py drops `roofline.traffic` to null when the library it
    times is another build than the one the counters were taken on."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:12]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-2000:])
        return obj

    if jobs:
        if verbose:
            print("[build] hipcc gfx950:", ", ".join(os.path.basename(s) for s, _ in jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or force or _stale(LIB, objs):
        # hipcc keeps the host stubs of __global__ kernels at default visibility whatever -fvisibility says: the version script
        # makes the dynamic symbol table exactly the C ABI (every entry point is named sed_*)
        vs = os.path.join(OBJ, "exports.map")
        with open(vs, "w") as fh:
            fh.write("{ global: sed_*; local: *; };\n")
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
