"""Phase stamps of the GRU forward recurrence: the -DGRU_STAMP build of sed_gru.hip (s_memtime at six points of a step
for the first and the last wave of workgroup 0, steps 16 .. 47) next to the unstamped kernel's time per step.
    ONLY=sed_gru.hip python tools/build_variant.py grustamp -DGRU_STAMP ; python tools/gru_stamps.py [tools/_libsed_grustamp.so]"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from desed_task_amd import _lib
B, T, H = 48, 156, 128
gi = torch.randn(B, T, 2, 3 * H, device="cuda")
whh = [torch.randn(3 * H, H, device="cuda") * 0.08 for _ in range(2)]
bhh = [torch.randn(3 * H, device="cuda") * 0.08 for _ in range(2)]
out = torch.empty(B, T, 2 * H, device="cuda")
saved = torch.empty(B, T, 2, 4, H, device="cuda")


def timed(lib, n=20):
    args = (gi.data_ptr(), whh[0].data_ptr(), whh[1].data_ptr(), bhh[0].data_ptr(), bhh[1].data_ptr(), out.data_ptr(), saved.data_ptr(), B, T, H,
            torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.call("sed_gru_fwd", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.call("sed_gru_fwd", *args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


_lib.use_library(None, is_emulator=False)
us = timed(_lib.get())
print("product kernel: %.1f us per launch = %.0f ns per step" % (us, us * 1e3 / T))
so = sys.argv[1] if len(sys.argv) > 1 else "tools/_libsed_grustamp.so"
_lib.use_library(so, is_emulator=False)
lib = _lib.get()
buf = torch.zeros(2 * 32 * 8, dtype=torch.int64, device="cuda")
dll = ctypes.CDLL(so)
assert dll.sed_gru_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
us_s = timed(lib, 5)
ts = buf.cpu().numpy().reshape(2, 32, 8)
print("stamped kernel: %.1f us per launch = %.0f ns per step" % (us_s, us_s * 1e3 / T))
names = ["top -> h + gate inputs landed (8 ds_read_b128 + 3 ds_read_b32)", "-> 48 packed FMAs done", "-> quarters summed (2 x 3 DPP adds)",
         "-> gates done (2 sigmoid, tanh, blend)", "-> results stored to LDS (landed)", "-> barrier passed"]
# clock: s_memtime counts at a fixed 100 MHz on gfx9 (REFCLK); report ticks and ns
for w, wn in ((0, "first wave"), (1, "last wave")):
    d = np.diff(ts[w, :, :7].astype(np.int64), axis=1)          # (32 steps, 6 phases)
    total = (ts[w, 1:, 0] - ts[w, :-1, 0]).astype(np.int64)
    inner = [s for s in range(32) if (16 + s) % 8 not in (0, 7)]         # steps that touch no chunk boundary
    print("== %s, steps 16..47 (ticks of s_memtime): step period median %d (chunk-interior %d)" % (
        wn, int(np.median(total)), int(np.median(total[[s for s in inner if s < 31]]))))
    for k, nme in enumerate(names):
        print("   %-72s median %5d   chunk-interior median %5d   max %5d" % (nme, int(np.median(d[:, k])), int(np.median(d[inner, k])), int(d[:, k].max())))
    print("   sum of the phase medians %d" % int(sum(np.median(d[inner, k]) for k in range(6))))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("(s_memtime ticks per step x steps vs the launch time gives the tick length: %.2f ns)" % (
    us_s * 1e3 / T / max(1.0, float(np.median((ts[0, 1:, 0] - ts[0, :-1, 0]).astype(np.int64))))))
