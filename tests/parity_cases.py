"""Parity cases shared by the CPU-emulator tests (small shapes) and the GPU tests (`-m gpu`, through the real
C-ABI library).  Every case compares the HIP path against the CPU oracle on the same seeded inputs."""
import numpy as np
import torch

from oracle import sed_oracle as O
from desed_task_amd import _lib
from desed_task_amd import features as Fh


def to(dev, *ts):
    r = [t.to(dev) for t in ts]
    return r[0] if len(r) == 1 else r


def case_mfma_selftest(dev):
    lib = _lib.get()
    g = torch.Generator().manual_seed(3)
    for shape, K in ((32, 8), (32, 64), (16, 12), (16, 64)):
        A = torch.randn(shape, K, generator=g)
        B = torch.randn(K, shape, generator=g)       # asymmetric B: catches transposed C maps
        C = torch.zeros(shape, shape)
        Ad, Bd, Cd = to(dev, A, B, C)
        lib.call("sed_selftest_mfma", Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), K, shape, _lib.stream_ptr(Ad))
        ref = (A.double() @ B.double()).float()
        err = (Cd.cpu() - ref).abs().max().item()
        assert err < 1e-4, (shape, K, err)


def make_mel():
    return Fh.MelSpectrogram(16000, 2048, 2048, 256, 0, 8000, 128, torch.hamming_window, {"periodic": False}, 1)


def case_mel(dev, batch=2, n_samples=256 * 12):
    audio = O.synth_audio(batch, n_samples, seed=1234)
    ref = O.mel_spectrogram(audio)
    mel = make_mel()
    got = mel(to(dev, audio))
    assert tuple(got.shape) == tuple(ref.shape)
    err = (got.cpu() - ref).abs().max().item()
    assert err < 2e-4 * ref.abs().max().item(), err
    a = O.scale_minmax(O.take_log(ref))
    b = Fh.minmax_scale(got, apply_log=True).cpu()
    assert (a - b).abs().max().item() < 1e-3          # north_star tolerance, scaler domain
    c = Fh.minmax_scale(Fh.take_log(got)).cpu()
    assert (a - c).abs().max().item() < 1e-3
    fused = mel.frames_major(to(dev, audio), apply_log=True).transpose(1, 2).cpu()
    assert (fused - O.take_log(ref)).abs().max().item() < 2e-2   # dB domain near the 1e-5 floor
    return got


def case_logscale_generic(dev):
    x = O.lcg_fill((3, 16, 9), 55, 30.0, -10.0)
    got, mm = Fh.minmax_scale(to(dev, x), return_minmax=True)
    np.testing.assert_allclose(got.cpu().numpy(), O.scale_minmax(x).numpy(), atol=1e-6)
    np.testing.assert_allclose(mm[:, 0].cpu().numpy(), x.amin((1, 2)).numpy())
    np.testing.assert_allclose(mm[:, 1].cpu().numpy(), x.amax((1, 2)).numpy())


def case_mixup_specaug(dev):
    x = O.lcg_fill((6, 8, 20), 5, 1.0, 1.5)
    y = (O.lcg_fill((6, 10, 7), 6, 0.5, 0.5) < 0.3).float()
    perm = torch.tensor([3, 0, 5, 1, 2, 4])
    c = 0.3721
    rx, ry = O.mixup_apply(x, y, c, perm)
    gx = Fh.mixup_(to(dev, x.clone()), perm, c, mode=0)
    gy = Fh.mixup_(to(dev, y.clone()), perm, c, mode=1)
    np.testing.assert_allclose(gx.cpu().numpy(), rx.numpy(), atol=1e-6)
    np.testing.assert_allclose(gy.cpu().numpy(), ry.numpy(), atol=1e-6)
    xin = O.lcg_fill((3, 16, 30), 9, 1.0)
    bounds = torch.tensor([[2, 5, 10, 13], [0, 0, 29, 30], [15, 16, 0, 4]], dtype=torch.int32)
    ref = O.specaug_apply(xin, (bounds[:, 0].long(), bounds[:, 1].long()), (bounds[:, 2].long(), bounds[:, 3].long()))
    got = Fh.specaug_apply(to(dev, xin), to(dev, bounds))
    assert torch.equal(got.cpu(), ref)
