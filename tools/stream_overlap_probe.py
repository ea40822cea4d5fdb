"""Do two HIP streams overlap kernels on this box?  Runs the GRU forward (96 workgroups) on 1 vs 2 streams."""
import sys, torch
sys.path.insert(0, '.')
from desed_task_amd import _lib
lib = _lib.get()
B, T, H = 48, 156, 128
dev = "cuda"
def mk():
    return dict(gi=torch.randn(B, T, 2, 3 * H, device=dev), w=[torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(2)],
                b=[torch.randn(3 * H, device=dev) * 0.05 for _ in range(2)], out=torch.empty(B, T, 2 * H, device=dev),
                sv=torch.empty(B, T, 2, 4, H, device=dev))
a, b = mk(), mk()
def run(d, stream):
    lib.call("sed_gru_fwd", d["gi"].data_ptr(), d["w"][0].data_ptr(), d["w"][1].data_ptr(), d["b"][0].data_ptr(), d["b"][1].data_ptr(),
             d["out"].data_ptr(), d["sv"].data_ptr(), B, T, H, stream.cuda_stream)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    import time; t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
seq = timed(lambda: (run(a, s1), run(b, s1)))
par = timed(lambda: (run(a, s1), run(b, s2)))
print("two GRU fwd launches: same stream %.1f us, two streams %.1f us" % (seq, par))
