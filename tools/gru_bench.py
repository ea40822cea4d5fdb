"""Times one BiGRU layer (forward and backward through ops.BiGRULayerFn) at B = 48, T = 156 for the two recurrent widths (diagnostics)."""
import sys
import torch
sys.path.insert(0, ".")
from desed_task_amd.ops import BiGRULayerFn
from desed_task_amd import _lib
if len(sys.argv) > 1:                 # A/B: another build of the C-ABI library (tools/_libsed_*.so)
    _lib.use_library(sys.argv[1], is_emulator=False)
lib = _lib.get(); orig = lib.call
B, T = 48, 156
for H, I in ((128, 128), (128, 256), (192, 128), (192, 384)):
    x = torch.randn(B, T, I, device="cuda", requires_grad=True)
    ws = []
    for _ in range(2):
        ws += [torch.randn(3 * H, I, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08, torch.zeros(3 * H, device="cuda"), torch.zeros(3 * H, device="cuda")]
    ws = [w.requires_grad_(True) for w in ws]
    rec = {}
    def timed(name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a); e1.record()
        rec.setdefault(name, []).append((e0, e1))
    for it in range(4):
        if it == 3:
            lib.call = timed
        out = BiGRULayerFn.apply(x, *ws)
        out.backward(torch.ones_like(out))
    torch.cuda.synchronize(); lib.call = orig
    print("H=%d I=%d:" % (H, I), {k: round(sum(a.elapsed_time(b) for a, b in v) * 1e3, 1) for k, v in rec.items()})
