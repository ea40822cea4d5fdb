"""Mel kernel alone, B = 48 clips of 10 s: variants (tuning key mel_wave: 0 = the default wave-per-frame kernel, 2 = the round-1..4
one-frame-per-workgroup kernel), HIP-event time per launch, agreement between the variants, and the error against an f64 FFT.
SED_PROBE_LIB=<library.so>: a variant build (tools/build_variant.py) instead of the product library."""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from tests import parity_cases as P
from desed_task_amd import _lib
if os.environ.get("SED_PROBE_LIB"):
    _lib.use_library(os.environ["SED_PROBE_LIB"], is_emulator=False)
mel = P.make_mel()
g = torch.Generator().manual_seed(3)
audio = (0.1 * torch.randn(48, 160000, generator=g)).cuda()
ref = None
for v in [int(a) for a in sys.argv[1:]] or [0, 2]:
    _lib.set_tuning("mel_wave", v)
    for _ in range(3): out = mel(audio)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = mel(audio)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    msg = ""
    if ref is None:
        ref = out.clone()
        x = audio[:2].double().cpu()
        xp = torch.nn.functional.pad(x[:, None], (1024, 1024), mode="reflect")[:, 0]
        fr = xp.unfold(1, 2048, 256) * torch.hamming_window(2048, periodic=False, dtype=torch.float64)
        mag = torch.fft.rfft(fr, dim=-1).abs()
        want = (mag @ mel.fb_dense.double().cpu())
        got = out[:2].transpose(1, 2).double().cpu()
        msg = " | vs f64: max rel-to-max err %.2e" % ((got - want).abs().max() / want.abs().max()).item()
    else:
        msg = " | vs first variant: max abs diff %.3e (max %.3e)" % ((out - ref).abs().max().item(), ref.abs().max().item())
    print("mel_wave=%d: %.1f us / launch -> %.0f GB/s algorithmic%s" % (v, ms * 1e3, 48 * 960512 / ms / 1e6, msg))
_lib.set_tuning("mel_wave", 0)
