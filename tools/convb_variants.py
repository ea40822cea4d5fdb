"""Timing ablation of the split-bf16 3x3 conv kernel (diagnostics).  Builds sed_conv_bf16.hip with -DCONVB_ABL=mask into
tools/_convb_v{mask}.so (1 = no weight-slab loads, 2 = no MFMAs, 4 = no patch loads, 8 = no per-tap barrier) and times
sed_conv3x3_bf16x3 at B = 48 on the recipe's layer shapes."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "desed_task_amd", "csrc")
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 9, 15]
for v in variants:
    so = os.path.join(HERE, "_convb_v%d.so" % v)
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "sed_conv_bf16.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC,
                               "-I", os.path.join(ROOT, "include"), "-DCONVB_ABL=%d" % v, os.path.join(CSRC, "sed_conv_bf16.hip"), "-o", so])
if not torch.cuda.is_available():
    sys.exit(0)
sys.path.insert(0, ROOT)
from desed_task_amd.ops import pack_conv_weights
P, I = ctypes.c_void_p, ctypes.c_int
for (CIN, COUT, F) in [(128, 128, 8), (64, 128, 16)]:
    B, T = 48, 156 if CIN >= 64 else 313
    x = torch.randn(B, T, F, CIN, device="cuda")
    w = torch.randn(COUT, CIN, 3, 3, device="cuda") * 0.03
    bias = torch.zeros(COUT, device="cuda")
    y = torch.empty(B, T, F, COUT, device="cuda")
    partial = torch.empty(8192 * 2 * COUT, device="cuda")
    (wf, wd), = pack_conv_weights([w], True, "bf16x3")
    st = torch.cuda.current_stream().cuda_stream
    for v in variants:
        lib = ctypes.CDLL(os.path.join(HERE, "_convb_v%d.so" % v))
        f = lib.sed_conv3x3_bf16x3
        f.argtypes = [P] * 5 + [I] * 5 + [P]
        a = (x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), partial.data_ptr(), B, T, F, CIN, COUT, st)
        for _ in range(3):
            assert f(*a) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*a)
        e1.record(); torch.cuda.synchronize()
        print("conv %3d->%3d F=%2d abl=%2d: %.1f us" % (CIN, COUT, F, v, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
