"""Tile-size sweep of the split-bf16 3x3 conv (sed_set_tuning overrides) over every forward / data-gradient shape of the recipe."""
import os, sys
sys.path.insert(0, '.')
import torch
from desed_task_amd import _lib
from desed_task_amd.ops import pack_conv_weights
lib = _lib.get()
B = 48
shapes = [(16, 32, 313, 64), (32, 64, 156, 32), (64, 128, 156, 16), (128, 128, 156, 8), (128, 128, 156, 4), (128, 128, 156, 2),
          (32, 16, 313, 64), (64, 32, 156, 32), (128, 64, 156, 16)]
for (CIN, COUT, T, F) in shapes:
    x = torch.randn(B, T, F, CIN, device="cuda")
    w = torch.randn(COUT, CIN, 3, 3, device="cuda") * 0.03
    bias = torch.zeros(COUT, device="cuda")
    y = torch.empty(B, T, F, COUT, device="cuda")
    partial = torch.empty(16384 * 2 * COUT, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for ck, mp in (("32", "default"), ("32", "128"), ("32", "256"), ("16", "64"), ("16", "128"), ("16", "256")):
        _lib.set_tuning("convb_ck", int(ck))            # the packing and the dispatch both read it
        (wf, wd), = pack_conv_weights([w], True, "bf16x3")
        if mp == "default":
            _lib.set_tuning("convb_mp", 0)
        else:
            _lib.set_tuning("convb_mp", int(mp))
        def run():
            return lib.value("sed_conv3x3_bf16x3", x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), partial.data_ptr(), B, T, F, CIN, COUT, st)
        if run() != 0:
            res.append("ck%s/%s: n/a" % (ck, mp)); continue
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append("ck%s/%s: %.1f" % (ck, mp, e0.elapsed_time(e1) / 20 * 1e3))
    print("conv %3d->%3d T=%d F=%2d  us  " % (CIN, COUT, T, F) + "  ".join(res), flush=True)
