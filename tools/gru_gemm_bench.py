"""Times the weight-gradient GEMM pair of a BiGRU layer (dW_ih[d] = dgi[d]^T x: M = 3H, N = I, K = B T, A and B both K-major) for
several split-K factors (diagnostics)."""
import sys
import torch
sys.path.insert(0, ".")
from desed_task_amd import _lib
lib = _lib.get()
st = torch.cuda.current_stream().cuda_stream
B, T, H = 48, 156, 128
BT = B * T
for I in (128, 256):
    dgi = torch.randn(BT, 2, 3 * H, device="cuda"); x = torch.randn(BT, I, device="cuda")
    dw0 = torch.zeros(3 * H, I, device="cuda"); dw1 = torch.zeros(3 * H, I, device="cuda")
    off = 3 * H * 4
    for split in (4, 8, 13, 20, 29, 32, 48):
        a = (dgi.data_ptr(), dgi.data_ptr() + off, x.data_ptr(), x.data_ptr(), None, None, dw0.data_ptr(), dw1.data_ptr(),
             3 * H, I, BT, 6 * H, I, I, 1, 0, split, 0, st)
        for _ in range(3):
            lib.call("sed_gemm_pair_bf16x3", *a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.call("sed_gemm_pair_bf16x3", *a)
        e1.record(); torch.cuda.synchronize()
        print("I=%d split=%2d: %.1f us" % (I, split, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
