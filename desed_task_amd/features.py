"""Mel front-end and spectrogram-domain ops on the MI355X kernels (K1-K5).

Mirrors what recipes/dcase2023_task4_baseline/local/sed_trainer.py builds from torchaudio:
`MelSpectrogram(...)` (:80-91), `take_log` (:253-264), and the TorchScaler instance/minmax call
inside `detect` (:266-267).  Shapes at the API are the reference's (B, n_mels, T); in HBM the data
is frame-major (B, T, n_mels) and handed out as a transposed view, so every kernel streams whole
frames and conv0 consumes it without a permute.
"""
import math
import os

import numpy as np
import torch

from . import _lib


def as_btf(x):
    """(B, F, T) reference-shaped tensor -> contiguous (B, T, F) tensor (no copy when x is our view)."""
    xt = x.transpose(1, 2)
    return xt if xt.is_contiguous() else xt.contiguous()


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """HTK triangular filters, norm=None, fp32 like torchaudio.functional.melscale_fbanks -> (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class MelSpectrogram(torch.nn.Module):
    """Drop-in for torchaudio.transforms.MelSpectrogram as configured by the DCASE recipes.

    Supported configuration (the one every recipe uses): n_fft == win_length == 2048, power == 1,
    center=True/reflect, HTK mel scale, norm=None, n_mels <= 128.  Anything else raises."""

    def __init__(self, sample_rate=16000, n_fft=2048, win_length=None, hop_length=256, f_min=0.0, f_max=None,
                 n_mels=128, window_fn=torch.hamming_window, wkwargs=None, power=1, **unsupported):
        super().__init__()
        if unsupported:
            raise NotImplementedError("MelSpectrogram options not supported by the HIP kernel: %s" % sorted(unsupported))
        win_length = n_fft if win_length is None else win_length
        if n_fft != 2048 or win_length != n_fft or power != 1 or n_mels > 128:
            raise NotImplementedError("HIP mel kernel supports n_fft=win_length=2048, power=1, n_mels<=128")
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        f_max = float(sample_rate // 2) if f_max is None else float(f_max)
        window = window_fn(n_fft, **(wkwargs or {})).float()
        n = np.arange(1024, dtype=np.float64)
        tw1024 = np.stack([np.cos(2 * np.pi * n / 1024), -np.sin(2 * np.pi * n / 1024)], 1).astype(np.float32)
        tw2048 = np.stack([np.cos(2 * np.pi * n / 2048), -np.sin(2 * np.pi * n / 2048)], 1).astype(np.float32)
        fb = melscale_fbanks_htk(n_fft // 2 + 1, float(f_min), f_max, n_mels, sample_rate)     # (1025, n_mels)
        nz = fb != 0
        start = torch.zeros(n_mels, dtype=torch.int32)
        length = torch.zeros(n_mels, dtype=torch.int32)
        for m in range(n_mels):
            idx = torch.nonzero(nz[:, m]).flatten()
            if idx.numel():
                start[m] = int(idx[0])
                length[m] = int(idx[-1]) - int(idx[0]) + 1
        stride = max(8, int(length.max()))
        w = torch.zeros(n_mels, stride)
        for m in range(n_mels):
            w[m, : int(length[m])] = fb[int(start[m]): int(start[m]) + int(length[m]), m]
        self.fb_stride = stride
        self.register_buffer("window", window, persistent=False)
        self.register_buffer("tw1024", torch.from_numpy(tw1024).contiguous(), persistent=False)
        self.register_buffer("tw2048", torch.from_numpy(tw2048).contiguous(), persistent=False)
        self.register_buffer("fb_start", start, persistent=False)
        self.register_buffer("fb_len", length, persistent=False)
        self.register_buffer("fb_w", w.contiguous(), persistent=False)
        self.register_buffer("fb_dense", fb.contiguous(), persistent=False)     # (n_freqs, n_mels): checkpoint compatibility only

    # Checkpoint compatibility with the reference's SEDTask4 state dict: torchaudio's MelSpectrogram owns two persistent buffers,
    # `spectrogram.window` and `mel_scale.fb`, which therefore sit in every Lightning checkpoint of the recipe under
    # `mel_spec.*` (train_sed.py:302 loads them strictly).  They are emitted on save and accepted (and ignored: both are functions
    # of the constructor arguments) on load.
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + "spectrogram.window"] = self.window if keep_vars else self.window.detach()
        destination[prefix + "mel_scale.fb"] = self.fb_dense if keep_vars else self.fb_dense.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in ("spectrogram.window", "mel_scale.fb"):
            state_dict.pop(prefix + k, None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    TAPS_FLOATS = (4 + 12) * 64 * 4     # sed_mel_taps / mel_wave_kernel: MEL_GA + MEL_GB groups of four taps per lane
    _taps = None

    def _tables_to(self, device):
        if self.window.device != device:
            self.to(device)

    def frames_major(self, audio, apply_log=False, out=None):
        """audio (B, N) -> (B, T, n_mels) contiguous (written into `out` when given: the pipelined front-end's feature buffer)."""
        if audio.dim() != 2:
            raise ValueError("audio must be (batch, samples)")
        audio = audio.float()
        if not audio.is_contiguous():
            audio = audio.contiguous()
        _lib.check_tensor(audio, "audio")
        self._tables_to(audio.device)
        B, N = audio.shape
        T = 1 + N // self.hop_length
        if out is None:
            out = torch.empty(B, T, self.n_mels, device=audio.device, dtype=torch.float32)
        elif tuple(out.shape) != (B, T, self.n_mels) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != audio.device:
            raise ValueError("frames_major: `out` must be a contiguous fp32 (B, T, n_mels) tensor on the audio's device")
        lib = _lib.get()
        if _lib.get_tuning("mel_wave") == 2 or os.environ.get("SED_MEL_WAVE") == "0" or _lib.get_tuning("mel_taps_mem"):
            # the round-1..4 kernel, one frame per 256-thread workgroup (141 - 150 us at B = 48): kept for A/B runs (`mel_wave` = 2 /
            # SED_MEL_WAVE=0) and for filterbanks whose taps do not fit the wave kernel's tables (`mel_taps_mem`)
            lib.call("sed_mel_fwd", audio.data_ptr(), out.data_ptr(), B, N, T, self.n_fft, self.hop_length, self.n_mels,
                     self.window.data_ptr(), self.tw1024.data_ptr(), self.tw2048.data_ptr(), self.fb_start.data_ptr(),
                     self.fb_len.data_ptr(), self.fb_w.data_ptr(), self.fb_stride, int(apply_log), _lib.stream_ptr(audio))
            return out
        # DEFAULT (round 6): one wave per frame and one frame per wave (`sed_mel_fwd_wave`).  Round 5's form of this kernel, in which a
        # wave transformed several frames per launch, returned a few wrong bins in single frames when replayed as a hipGraph node beside
        # the split-bf16 GEMM; with exactly one frame per wave the fault has never been seen (tests/test_gpu_parity.py::
        # test_mel_in_graph_beside_{tails,gemm}: 3 000 replays each per GPU session; profiles/r06_mel_mechanism.md)
        taps = self._taps
        if taps is None or taps.device != audio.device:
            # the filterbank in the wave kernel's LDS layout: one tiny launch per device, outside any captured step
            taps = torch.zeros(self.TAPS_FLOATS, device=audio.device, dtype=torch.float32)
            lib.call("sed_mel_taps", self.fb_start.data_ptr(), self.fb_len.data_ptr(), self.fb_w.data_ptr(), self.fb_stride,
                     self.n_mels, taps.data_ptr(), _lib.stream_ptr(audio))
            self._taps = taps
        lib.call("sed_mel_fwd_wave", audio.data_ptr(), out.data_ptr(), B, N, T, self.n_fft, self.hop_length, self.n_mels,
                 self.window.data_ptr(), self.tw1024.data_ptr(), self.tw2048.data_ptr(), self.fb_start.data_ptr(),
                 self.fb_len.data_ptr(), self.fb_w.data_ptr(), self.fb_stride, taps.data_ptr(), int(apply_log), _lib.stream_ptr(audio))
        return out

    def forward(self, audio):
        return self.frames_major(audio).transpose(1, 2)          # (B, n_mels, T) view


def take_log(mels):
    """SEDTask4.take_log: 20*log10(clamp(x, 1e-5)).clamp(-50, 80)."""
    if mels.dim() == 3 and not mels.is_contiguous():
        xt = as_btf(mels)
        _lib.check_tensor(xt, "mels")
        y = torch.empty_like(xt)
        _lib.get().call("sed_take_log", xt.data_ptr(), y.data_ptr(), xt.numel(), _lib.stream_ptr(xt))
        return y.transpose(1, 2)
    x = mels.contiguous()
    _lib.check_tensor(x, "mels")
    y = torch.empty_like(x)
    _lib.get().call("sed_take_log", x.data_ptr(), y.data_ptr(), x.numel(), _lib.stream_ptr(x))
    return y


def minmax_scale(x, eps=1e-8, apply_log=False, return_minmax=False, out=None):
    """Instance min-max scaling over all non-batch dims to [-1, 1]; optionally fused with take_log.
    Works on any layout whose clips are contiguous blocks (our (B,F,T) views included).
    out: a tensor of x's shape AND strides to write into (the pipelined front half's hand-over buffer) instead of a fresh one."""
    if x.dim() == 3 and not x.is_contiguous():
        xt = as_btf(x)
        view_back = True
    else:
        xt = x.contiguous()
        view_back = False
    _lib.check_tensor(xt, "features")
    B = xt.shape[0]
    L = xt.numel() // max(B, 1)
    if out is None:
        out = torch.empty_like(xt)
    else:
        if out.shape != x.shape or out.stride() != x.stride() or out.dtype != xt.dtype or out.device != xt.device:
            raise ValueError("minmax_scale: `out` must have the input's shape, strides, dtype and device")
        out = out.transpose(1, 2) if view_back else out
        if not out.is_contiguous():
            raise ValueError("minmax_scale: `out` must be dense in the input's layout")
    partial = torch.empty(B * 64, device=xt.device, dtype=torch.float32)
    mm = torch.empty(B, 2, device=xt.device, dtype=torch.float32) if return_minmax else None
    _lib.get().call("sed_logscale_fwd", xt.data_ptr(), out.data_ptr(), out.data_ptr(), partial.data_ptr(),
                    mm.data_ptr() if mm is not None else None, B, L, int(apply_log), float(eps), _lib.stream_ptr(xt))
    res = out.transpose(1, 2) if view_back else out
    return (res, mm) if return_minmax else res


def mixup_(data, perm, c, mode=0, c_dev=None, perm_dev=None):
    """In-place mixup of a group: data[i] <- c*data[i] + (1-c)*data[perm[i]] (perm: int tensor on any device).
    `data` must be a batch-major slice whose clips are contiguous blocks.
    c_dev / perm_dev: device addresses of the coefficient and the int32 permutation (graph.DynArgs) instead of c / perm."""
    n = data.shape[0]
    if n == 0:
        return data
    base = data
    if not data.is_contiguous():
        if data.dim() == 3 and data.transpose(1, 2).is_contiguous():
            base = data.transpose(1, 2)
        else:
            raise RuntimeError("mixup_: clips must be contiguous blocks")
    _lib.check_tensor(base, "mixup data")
    L = base.numel() // n
    tmp = torch.empty_like(base)
    if perm_dev is None:
        # blocking on purpose: the int32 conversion is a TEMPORARY pageable host tensor -- an asynchronous copy out of it reads
        # freed memory once the GPU lags behind the host (seen with two ranks time-slicing one GPU)
        perm_d = perm.to(torch.int32).to(base.device)
        perm_dev = perm_d.data_ptr()
    _lib.get().call("sed_mixup", base.data_ptr(), tmp.data_ptr(), perm_dev, float(np.float32(c)),
                    float(np.float32(1.0 - c)), n, L, int(mode), c_dev, _lib.stream_ptr(base))
    return data


MIXUP_MULTI_MAX_JOBS, MIXUP_MULTI_MAX_CLIPS = 8, 32


def _clip_major(data):
    """The tensor whose clips are contiguous blocks: `data` itself or its (B, T, F) transpose."""
    if data.is_contiguous():
        return data
    if data.dim() == 3 and data.transpose(1, 2).is_contiguous():
        return data.transpose(1, 2)
    raise RuntimeError("mixup_: clips must be contiguous blocks")


def mixup_multi_(jobs):
    """Several in-place mixups in ONE launch and without scratch copies (sed_mixup_multi).  jobs: [(data, perm, c, mode, c_dev,
    perm_dev)] as for mixup_(); groups of more than 32 clips (or more than 8 jobs) fall back to one mixup_() launch per job.
    The jobs of one call must be different tensors (mixing the same labels twice needs two calls: the second reads the first's
    result)."""
    import ctypes
    import struct
    jobs = [j for j in jobs if j[0].shape[0] > 0]
    if not jobs:
        return
    if len(jobs) > MIXUP_MULTI_MAX_JOBS or any(j[0].shape[0] > MIXUP_MULTI_MAX_CLIPS for j in jobs):
        for data, perm, c, mode, c_dev, perm_dev in jobs:
            mixup_(data, perm, c, mode=mode, c_dev=c_dev, perm_dev=perm_dev)
        return
    # host permutations (eager path): every DISTINCT permutation of the call goes up in ONE blocking copy (a group's features and
    # labels share theirs; the 2023 step used to pay four synchronous uploads here, the 2024 step six)
    perms, slot = [], {}
    for _, perm, _, _, _, perm_dev in jobs:
        if perm_dev is None and id(perm) not in slot:
            slot[id(perm)] = sum(p.numel() for p in perms)
            perms.append(perm.reshape(-1).to(torch.int32))
    perm_all = None
    if perms:
        # blocking on purpose (see mixup_): the concatenation is a temporary pageable host tensor
        perm_all = torch.cat(perms).to(_clip_major(jobs[0][0]).device)
    rec = []
    for data, perm, c, mode, c_dev, perm_dev in jobs:
        base = _clip_major(data)
        _lib.check_tensor(base, "mixup data")
        n = base.shape[0]
        if perm_dev is None:
            perm_dev = perm_all.data_ptr() + 4 * slot[id(perm)]
        cf, of = np.float32(c), np.float32(1.0 - c)
        bits = lambda v: struct.unpack("<I", struct.pack("<f", float(v)))[0]      # noqa: E731
        rec += [base.data_ptr(), perm_dev, c_dev or 0, bits(cf), bits(of), n, base.numel() // n, int(mode)]
    arr = (ctypes.c_longlong * len(rec))(*rec)
    _lib.get().call("sed_mixup_multi", ctypes.addressof(arr), len(jobs), _lib.stream_ptr(_clip_major(jobs[0][0])))


def specaug_bounds(batch, n_freq, n_time, f_l, f_p, t_l, t_p, device, iid_masks=True, generator=None, seed=None):
    """Draws of torchaudio's mask_along_axis(_iid) for CRNN.apply_specaugment -> (B,4) int32 [f0,f1,t0,t1).
    seed None: two torch.rand calls (frequency axis first, as in the reference) feed ONE kernel that does the reference's float32
    arithmetic; the tensor-op version of this function cost ~25 single-element launches per model call.
    seed (int, or a graph.DynSeed whose value lives in device memory): the uniforms come from the kernels' counter-based generator
    keyed by that seed -- no launch besides the bounds kernel itself (what the training step uses)."""
    n = batch if iid_masks else 1
    if seed is not None:
        out = torch.empty(batch, 4, dtype=torch.int32, device=device)
        if batch == 0:
            return out
        _lib.check_tensor(out, "specaug bounds")
        params = [min(cap, int(axis_len * p)) for cap, p, axis_len in ((f_l, f_p, n_freq), (t_l, t_p, n_time))]
        _lib.get().call("sed_specaug_bounds_seeded", out.data_ptr(), batch, n, params[0], n_freq, params[1], n_time,
                        int(seed) & 0xFFFFFFFF, getattr(seed, "dev", None), _lib.stream_ptr(out))
        return out
    params, us = [], []
    for cap, p, axis_len in ((f_l, f_p, n_freq), (t_l, t_p, n_time)):
        mask_param = min(cap, int(axis_len * p))
        params.append(mask_param)
        us.append(torch.rand(2, n, device=device, generator=generator) if mask_param >= 1 else None)
    out = torch.empty(batch, 4, dtype=torch.int32, device=device)
    if batch == 0:
        return out
    _lib.check_tensor(out, "specaug bounds")
    _lib.get().call("sed_specaug_bounds", us[0].data_ptr() if us[0] is not None else None,
                    us[1].data_ptr() if us[1] is not None else None, out.data_ptr(), batch, n, params[0], n_freq, params[1], n_time,
                    _lib.stream_ptr(out))
    return out


def specaug_request(batch, n_freq, n_time, f_l, f_p, t_l, t_p, iid_masks, seed):
    """The parameters of a seeded specaug_bounds() draw, for a consumer that makes it inside another launch (CNN prologue), or None
    when neither mask can be longer than 0."""
    params = [min(cap, int(axis_len * p)) for cap, p, axis_len in ((f_l, f_p, n_freq), (t_l, t_p, n_time))]
    if params[0] < 1 and params[1] < 1:
        return None
    return dict(n=batch if iid_masks else 1, f_param=params[0], n_freq=n_freq, t_param=params[1], n_time=n_time, seed=seed)


def specaug_bounds_from_request(batch, req, device):
    """The stand-alone launch for a specaug_request (same arithmetic, same bits)."""
    out = torch.empty(batch, 4, dtype=torch.int32, device=device)
    if batch == 0:
        return out
    _lib.check_tensor(out, "specaug bounds")
    seed = req["seed"]
    _lib.get().call("sed_specaug_bounds_seeded", out.data_ptr(), batch, req["n"], req["f_param"], req["n_freq"], req["t_param"],
                    req["n_time"], int(seed) & 0xFFFFFFFF, getattr(seed, "dev", None), _lib.stream_ptr(out))
    return out


def weak_labels(labels):
    """(labels (n, NC, T).sum(-1) > 0).float() in one launch (sed_trainer.py:292)."""
    labels = labels.contiguous().float()
    n, NC, T = labels.shape
    out = torch.empty(n, NC, dtype=torch.float32, device=labels.device)
    if n == 0:
        return out
    _lib.check_tensor(labels, "labels")
    _lib.get().call("sed_weak_labels", labels.data_ptr(), out.data_ptr(), n, NC, T, _lib.stream_ptr(labels))
    return out


def specaug_apply(x, bounds):
    """x (B, F, T) reference-shaped -> masked copy (same view convention)."""
    xt = as_btf(x)
    _lib.check_tensor(xt, "specaug input")
    y = torch.empty_like(xt)
    B, T, Fq = xt.shape
    _lib.get().call("sed_specaug", xt.data_ptr(), y.data_ptr(), bounds.data_ptr(), B, T, Fq, _lib.stream_ptr(xt))
    return y.transpose(1, 2)
