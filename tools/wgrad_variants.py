"""Timing ablation of the split-bf16 all-taps weight gradient (diagnostics): builds sed_conv.hip with -DWGN_ABL=mask into
tools/_wgn_v{mask}.so (1 = no MFMAs, 2 = no operand reads / MFMAs, 4 = no staging, 8 = no global loads) and times
sed_conv_wgrad_bf16x3 on the two narrow layers at B = 48."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "desed_task_amd", "csrc")
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 14]
for v in variants:
    so = os.path.join(HERE, "_wgn_v%d.so" % v)
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "sed_conv.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC,
                               "-I", os.path.join(ROOT, "include"), "-DWGN_ABL=%d" % v, os.path.join(CSRC, "sed_conv.hip"),
                               os.path.join(CSRC, "sed_selftest.hip"), "-o", so])
if not torch.cuda.is_available():
    sys.exit(0)
P, I = ctypes.c_void_p, ctypes.c_int
st = torch.cuda.current_stream().cuda_stream
for (T, F, CIN, COUT) in [(313, 64, 16, 32), (156, 32, 32, 64)]:
    B = 48
    x = torch.randn(B, T, F, CIN, device="cuda"); dy = torch.randn(B, T, F, COUT, device="cuda")
    dW = torch.empty(COUT, CIN, 3, 3, device="cuda")
    for v in variants:
        lib = ctypes.CDLL(os.path.join(HERE, "_wgn_v%d.so" % v))
        lib.sed_set_tuning(7, int(os.environ.get("WGN_CAP", "0")))
        lib.sed_conv_wgrad_scratch_floats.restype = ctypes.c_longlong
        scr = torch.empty(int(lib.sed_conv_wgrad_scratch_floats(B, T, F, CIN, COUT)), device="cuda")
        f = lib.sed_conv_wgrad_bf16x3
        f.argtypes = [P] * 4 + [I] * 5 + [P]
        a = (x.data_ptr(), dy.data_ptr(), scr.data_ptr(), dW.data_ptr(), B, T, F, CIN, COUT, st)
        for _ in range(3):
            assert f(*a) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*a)
        e1.record(); torch.cuda.synchronize()
        print("%d->%d cap=%s abl=%2d: %.1f us (incl. reduce)" % (CIN, COUT, os.environ.get("WGN_CAP", "0"), v, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
