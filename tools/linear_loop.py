"""One tile Linear (QKV shape) in a loop for power / clock sampling.  ZERO=1: all-zero operands."""
import os, sys, torch
sys.path.insert(0, ".")
from desed_task_amd import _lib
lib = _lib.get()
M, N, K = 23808, 2304, 768
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g); W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
if os.environ.get("ZERO"): A.zero_(); W.zero_()
C = torch.empty(M, N, device="cuda")
At = torch.empty(2 * ((M + 255) // 256) * 256 * K, dtype=torch.int16, device="cuda"); Wt = torch.empty(2 * N * K, dtype=torch.int16, device="cuda")
st = _lib.stream_ptr(A)
lib.call("sed_split_tiles_bf16x3", A.data_ptr(), At.data_ptr(), M, K, st); lib.call("sed_split_tiles_bf16x3", W.data_ptr(), Wt.data_ptr(), N, K, st)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), None, C.data_ptr(), M, N, K, 0, st)
e1.record(); torch.cuda.synchronize()
print("%d launches, %.1f us each" % (n, e0.elapsed_time(e1) / n * 1e3))
