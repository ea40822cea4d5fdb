"""Times the fused first block (sed_conv0_fwd statistics pass, sed_block0_fwd, sed_block0_bwd) at B = 48 through ConvBlockFn (diagnostics)."""
import sys
import torch
sys.path.insert(0, ".")
from desed_task_amd.ops import ConvBlockFn
from desed_task_amd import _lib
import os
for kv in sys.argv[1:]:                 # lib=<path>: time another build of the library (same-box A/B)
    if kv.startswith("lib="):
        _lib.use_library(os.path.abspath(kv[4:]), is_emulator=False)
lib = _lib.get(); orig = lib.call; rec = {}
for kv in sys.argv[1:]:                 # e.g. block0_bwd_v1=1 glu_grid_cap=640
    if kv.startswith("lib="):
        continue
    key, v = kv.split("="); _lib.set_tuning(key, int(v))
B, T, F = 48, 626, 128
x = torch.randn(B, T, F, device="cuda", requires_grad=False)
w = (torch.randn(16, 1, 3, 3, device="cuda") * 0.3).requires_grad_(True)
ps = [torch.randn(16, device="cuda").requires_grad_(True) for _ in range(3)] + [(torch.randn(16, 16, device="cuda") * 0.2).requires_grad_(True), torch.randn(16, device="cuda").requires_grad_(True)]
rm, rv = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda")
cfg = dict(pool=(2, 2), bn_training=True, dropout_p=0.5, apply_dropout=True, seed=7, bounds=None, update_running=True, conv_precision="bf16x3")
def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record()
    rec.setdefault(name, []).append((e0, e1))
for it in range(8):
    if it == 3:
        lib.call = timed
    out = ConvBlockFn.apply(x, w, ps[0], ps[1], ps[2], ps[3], ps[4], rm, rv, dict(cfg))
    out.backward(torch.ones_like(out))
torch.cuda.synchronize(); lib.call = orig
print(" ".join(sys.argv[1:]) or "default", {k: round(sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] * 1e3, 1) for k, v in rec.items()})
