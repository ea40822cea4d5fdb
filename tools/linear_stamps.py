"""Phase stamps of linear_dma_kernel (round 6): the -DT_STAMP build of sed_gemm_bf16.hip (s_memtime at ten seams of a K tile's two phases,
wave 0 = group 0 and wave 4 = group 1 of workgroup 64, K tiles 8 .. 23) next to the unstamped kernel's time.
    ONLY=sed_gemm_bf16.hip python tools/build_variant.py tstamp -DT_STAMP ; python tools/linear_stamps.py [shape]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from desed_task_amd import _lib
shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
M = 23808
N, K, act = {"qkv": (2304, 768, 0), "out": (768, 768, 0), "fc1": (3072, 768, 1), "fc2": (768, 3072, 0)}[shape]
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
b = torch.randn(N, device="cuda", generator=g); C = torch.empty(M, N, device="cuda")
At = torch.empty(2 * ((M + 255) // 256) * 256 * K, dtype=torch.int16, device="cuda"); Wt = torch.empty(2 * N * K, dtype=torch.int16, device="cuda")


def timed(lib, n=10):
    st = _lib.stream_ptr(A)
    lib.call("sed_split_tiles_bf16x3", A.data_ptr(), At.data_ptr(), M, K, st)
    lib.call("sed_split_tiles_bf16x3", W.data_ptr(), Wt.data_ptr(), N, K, st)
    run = lambda: lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, act, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if os.environ.get("FORM"): _lib.set_tuning("linear_tiles", int(os.environ["FORM"]))
us = timed(_lib.get())
print("%s: product kernel %.1f us per launch" % (shape, us))
so = os.environ.get("STAMP_LIB", "tools/_libsed_tstamp.so")
_lib.use_library(so, is_emulator=False)
buf = torch.zeros(3 * 16 * 16, dtype=torch.int64, device="cuda")
if os.environ.get("FORM"): _lib.set_tuning("linear_tiles", int(os.environ["FORM"]))      # FORM=5: the loader-wave kernel
assert ctypes.CDLL(so).sed_linear_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
us_s = timed(_lib.get(), 3)
print("stamped kernel %.1f us per launch" % us_s)
ts = buf.cpu().numpy().reshape(3, 16, 16)[:2, :, :10].astype(np.int64)
names = ["phase 0: DMA A issued, B + A fragments landed", "first barrier passed", "12 MFMAs issued", "second barrier passed",
         "phase 1: DMA W issued, A fragments landed", "vmcnt(8): tile kt + 1 landed", "first barrier passed", "12 MFMAs issued", "second barrier passed"]
for grp in (0, 1):
    d = np.diff(ts[grp], axis=1)
    period = np.diff(ts[grp, :, 0])
    print("== group %d (wave %d), K tiles 8 .. 23, s_memtime ticks: K-tile period median %d (min %d, max %d)" % (grp, 4 * grp, np.median(period), period.min(), period.max()))
    for i, nme in enumerate(names):
        print("   %-50s median %5d   min %5d   max %5d" % (nme, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
    print("   sum of medians %d" % np.median(d, axis=0).sum())
full = buf.cpu().numpy().reshape(3, 16, 16).astype(np.int64)
for grp in (0, 1):
    print("group %d: phase 0 start -> reads landed %d, -> DMA A issued %d | phase 1 start -> reads landed %d, -> DMA W issued %d, -> cursor advanced %d" % (
        grp, np.median(full[grp, :, 10] - full[grp, :, 0]), np.median(full[grp, :, 1] - full[grp, :, 10]), np.median(full[grp, :, 11] - full[grp, :, 4]),
        np.median(full[grp, :, 12] - full[grp, :, 11]), np.median(full[grp, :, 5] - full[grp, :, 12])))
print("group 1 minus group 0 at the phase-0 start: median %d ticks" % np.median(ts[1, :, 0] - ts[0, :, 0]))

if os.environ.get("FORM") == "5":
    ld = full[2, :, :10]
    d = np.diff(ld, axis=1)
    print("== loader wave 8: ticks between its stamps (issue 4 pieces | barrier) x 4, the 4th incl. the vmcnt wait")
    for i, nme in enumerate(["issue q0", "barrier", "issue q1", "barrier", "issue q2", "barrier", "issue q3 + cursor", "vmcnt(32)", "barrier"]):
        print("   %-22s median %5d  min %5d  max %5d" % (nme, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
    print("   step period median %d" % np.median(np.diff(ld[:, 0])))
