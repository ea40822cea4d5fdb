"""Micro-benchmark of the 3x3 conv kernels at a production shape (diagnostics / PMC profiling target)."""
import sys, os
sys.path.insert(0, '.')
import torch
from desed_task_amd import _lib
from desed_task_amd.ops import pack_conv_weights
if os.environ.get("SED_LIB"): _lib.use_library(os.environ["SED_LIB"], is_emulator=False)       # a tools/build_variant.py build
lib = _lib.get()
B, T, F, CIN, COUT = 48, 156, int(os.environ.get("F", "8")), int(os.environ.get("CIN", "128")), int(os.environ.get("COUT", "128"))
x = torch.randn(B, T, F, CIN, device="cuda")
w = torch.randn(COUT, CIN, 3, 3, device="cuda") * 0.03
bias = torch.zeros(COUT, device="cuda")
y = torch.empty(B, T, F, COUT, device="cuda")
nblk = 4 * lib.value("sed_conv_fwd_blocks", B, T, F, CIN, COUT)
partial = torch.empty(nblk * 2 * COUT, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode in ("f32", "bf16x3"):
    (wf, wd), = pack_conv_weights([w], True, "f32" if mode == "f32" else "bf16x3")
    entry = {"f32": "sed_conv3x3", "bf16x3": "sed_conv3x3_bf16x3"}[mode]
    def run():
        lib.call(entry, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), partial.data_ptr(), B, T, F, CIN, COUT, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("%s conv %d->%d F=%d: %.1f us, %.1f TFLOP/s algorithmic" % (mode, CIN, COUT, F, us, 2.0 * B * T * F * 9 * CIN * COUT / us / 1e6), flush=True)
