"""Times sed_conv_wgrad_bf16x3 on the recipe's narrow layers at B = 48 (diagnostics): the split-bf16 all-taps kernel
(sed_set_tuning key wgrad_narrow = 0) against the exact-f32 one (1), plus the wide layers for reference."""
import sys
import torch
sys.path.insert(0, ".")
from desed_task_amd import _lib
lib = _lib.get()
st = torch.cuda.current_stream().cuda_stream
for (T, F, CIN, COUT) in [(313, 64, 16, 32), (156, 32, 32, 64), (156, 16, 64, 128), (156, 8, 128, 128), (156, 4, 128, 128), (156, 2, 128, 128)]:
    B = 48
    x = torch.randn(B, T, F, CIN, device="cuda"); dy = torch.randn(B, T, F, COUT, device="cuda")
    dW = torch.empty(COUT, CIN, 3, 3, device="cuda")
    ref = None
    for narrow in (0, 1):
        _lib.set_tuning("wgrad_narrow", narrow)
        _lib.set_tuning("wgrad_wide", narrow)            # 1 = one tap per workgroup (the round-1 kernel)
        scr = torch.empty(int(lib.value("sed_conv_wgrad_scratch_floats", B, T, F, CIN, COUT)), device="cuda")
        args = (x.data_ptr(), dy.data_ptr(), scr.data_ptr(), dW.data_ptr(), B, T, F, CIN, COUT, st)
        for _ in range(3):
            lib.call("sed_conv_wgrad_bf16x3", *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.call("sed_conv_wgrad_bf16x3", *args)
        e1.record(); torch.cuda.synchronize()
        if ref is None:
            ref = dW.clone()
        print("T=%d F=%d %d->%d narrow=%d: %.1f us (incl. reduce)  max|diff vs first|/max = %.2e" %
              (T, F, CIN, COUT, narrow, e0.elapsed_time(e1) / 20 * 1e3, float((dW - ref).abs().max() / ref.abs().max())), flush=True)
_lib.set_tuning("wgrad_narrow", 0)
_lib.set_tuning("wgrad_wide", 0)
