"""The BEATs encoder's Linear shapes alone (M = 23 808 tokens): packed-weight kernel vs the generic split-bf16 GEMM, HIP-event time per launch,
algorithmic TFLOP/s and the fraction of the 833 TFLOP/s that three bf16 MFMAs per product allow.  python tools/linear_bench.py [packed|generic]"""
import os, sys, torch
sys.path.insert(0, '.')
from desed_task_amd import _lib
import os
if os.environ.get('SED_LIB'): _lib.use_library(os.environ['SED_LIB'], is_emulator=False)       # a tools/build_variant.py build
lib = _lib.get()
shape_only = os.environ.get("SHAPE")                # SHAPE=qkv: one layer only (PMC runs)
which = sys.argv[1:] or ["tiles", "tiles+split", "packed", "generic"]
M = 23808
g = torch.Generator(device="cuda").manual_seed(1)
for (N, K, act, name) in ((2304, 768, 0, "qkv"), (768, 768, 0, "out"), (3072, 768, 1, "fc1+gelu"), (768, 3072, 0, "fc2")):
    if shape_only and name != shape_only: continue
    A = torch.randn(M + 8, K + 64, device="cuda", generator=g)[:M, :K].contiguous() if not os.environ.get('PAD') else torch.randn(M + 8, K + 64, device="cuda", generator=g)   # PAD=1: the -DPP_DIAG=128/256 timing builds read a padded pitch
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    if os.environ.get('ZERO'): A.zero_(); W.zero_()          # ZERO=1: all-zero operands (how much of the time is the power budget?)
    C = torch.empty(M, N, device="cuda")
    Wp = torch.zeros(2 * N * (K + 128), dtype=torch.int16, device="cuda")
    st = _lib.stream_ptr(A)
    lib.call("sed_pack_weights_bf16x3", W.data_ptr(), Wp.data_ptr(), N, K, st)
    At = torch.empty(2 * ((M + 255) // 256) * 256 * K, dtype=torch.int16, device="cuda")
    Wt = torch.empty(2 * N * K, dtype=torch.int16, device="cuda")
    lib.call("sed_split_tiles_bf16x3", W.data_ptr(), Wt.data_ptr(), N, K, st)
    lib.call("sed_split_tiles_bf16x3", A.data_ptr(), At.data_ptr(), M, K, st)
    for kind in which:
        def run():
            if kind in ("tiles", "tiles-noskew", "tiles-ldr"):         # the GEMM alone on pre-split images (the producers write them); "tiles+split" adds the activation's split pass
                lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, act, st)
                return
            if kind == "tiles+split":
                lib.call("sed_split_tiles_bf16x3", A.data_ptr(), At.data_ptr(), M, K, st)
                lib.call("sed_linear_tiles_bf16x3", At.data_ptr(), Wt.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, act, st)
                return
            if kind == "packed":
                lib.call("sed_linear_packed_bf16x3", A.data_ptr(), Wp.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, act, st)
            else:
                lib.call("sed_linear_bf16x3", A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr(), M, N, K, act, st)
        _lib.set_tuning("linear_tiles", {"tiles-noskew": 3, "tiles-ldr": 5}.get(kind, 0))      # 3: no start skew
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = 2.0 * M * N * K / ms / 1e9
        _lib.set_tuning("linear_tiles", 0)
        print("%-9s %-8s N %4d K %4d: %7.1f us  %6.1f TFLOP/s  = %.3f of 833" % (name, kind, N, K, ms * 1e3, tf, tf / 833.3))
