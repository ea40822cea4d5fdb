"""Cycles per K step of the tile Linear's main loop, without stamps: ONE 256 x 256 tile per CU (M = 65 536, N = 256) and a long K
(6 144 = 384 steps), so that launch time / 384 is the step period (prologue + epilogue < 2 %).  forms: 0 = the eight-wave kernel (shipped), 5 = the loader-wave kernel.
    python tools/linear_period.py"""
import os, sys, torch
sys.path.insert(0, ".")
from desed_task_amd import _lib
if os.environ.get("SED_LIB"): _lib.use_library(os.environ["SED_LIB"], is_emulator=False)
lib = _lib.get()
M, N, K = 65536, 256, 6144
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
if os.environ.get("ZERO"): A.zero_(); W.zero_()
C = torch.empty(M, N, device="cuda")
At = torch.empty(2 * M * K, dtype=torch.int16, device="cuda"); Wt = torch.empty(2 * N * K, dtype=torch.int16, device="cuda")
st = _lib.stream_ptr(A)
lib.call("sed_split_tiles_bf16x3", A.data_ptr(), At.data_ptr(), M, K, st)
lib.call("sed_split_tiles_bf16x3", W.data_ptr(), Wt.data_ptr(), N, K, st)
for form in (0, 5):
    _lib.set_tuning("linear_tiles", form)
    run = lambda: lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), None, C.data_ptr(), M, N, K, 0, st)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    print("form %d: %.1f us per launch = %.0f ns per K step (48 MFMAs per SIMD = 1 536 matrix-pipe cycles)  %.1f TFLOP/s = %.3f of 833" % (form, us, us * 1e3 / (K // 16), tf, tf / 833.3))
_lib.set_tuning("linear_tiles", 0)
