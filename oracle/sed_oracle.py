"""CPU oracle for the mel + CRNN mean-teacher training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32, unfused) restatement of the reference's
algorithm for the hot path named in BASELINE.json.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``desed_task_amd/`` imports or calls it.

Pinning status (see DESIGN.md "Oracle"):
  * CRNN forward/backward, TorchScaler, mixup, ExponentialWarmup, update_ema and the
    SEDTask4.training_step scalars are pinned against the reference itself, imported in the
    build container under stubs (tests/golden/make_golden.py -> tests/golden/*.npz,
    checked by tests/test_oracle_golden.py).
  * The mel front-end, AmplitudeToDB and the SpecAugment axis mask live in torchaudio,
    which is neither vendored in the reference nor installed here: **parity unpinned** by
    the reference for those three.  They follow torchaudio's published semantics and are
    cross-checked against an independent float64 numpy implementation (fixture G1).

All tensors are (B, n_mels, T) like the reference unless stated otherwise.
Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------
# configuration of the 2023 recipe (recipes/dcase2023_task4_baseline/confs/default.yaml)
# ----------------------------------------------------------------------------------
FEATS = dict(n_mels=128, n_window=2048, hop_length=256, sample_rate=16000, f_min=0, f_max=8000)
NB_FILTERS = (16, 32, 64, 128, 128, 128, 128)
POOLING = ((2, 2), (2, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2))
BN_EPS = 1e-3        # desed_task/nnet/CNN.py:76
BN_MOMENTUM = 0.99   # desed_task/nnet/CNN.py:76 (torch semantics: weight of the NEW batch)


# ----------------------------------------------------------------------------------
# a1: mel front-end  (recipes/.../local/sed_trainer.py:80-91,282 -> torchaudio MelSpectrogram)
# ----------------------------------------------------------------------------------
def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs=1025, f_min=0.0, f_max=8000.0, n_mels=128, sample_rate=16000) -> torch.Tensor:
    """HTK triangular filterbank, norm=None, fp32 arithmetic like torchaudio.functional.melscale_fbanks.
    Returns (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # (n_freqs, n_mels+2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_spectrogram(audio: torch.Tensor, feats: dict = FEATS) -> torch.Tensor:
    """(B, N) waveform -> (B, n_mels, 1 + N // hop) linear-magnitude mel (power=1)."""
    n_fft, hop = feats["n_window"], feats["hop_length"]
    window = torch.hamming_window(n_fft, periodic=False, dtype=audio.dtype)
    spec = torch.stft(audio, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
    fb = mel_filterbank(n_fft // 2 + 1, float(feats["f_min"]), float(feats["f_max"]), feats["n_mels"],
                        feats["sample_rate"]).to(audio.dtype)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


# ----------------------------------------------------------------------------------
# a3: take_log (sed_trainer.py:253-264).  AmplitudeToDB(stype="amplitude") with amin patched to
# 1e-5 after construction: db_multiplier stays log10(max(1e-10, 1.0)) = 0.
# ----------------------------------------------------------------------------------
def take_log(mels: torch.Tensor) -> torch.Tensor:
    x_db = 20.0 * torch.log10(torch.clamp(mels, min=1e-5))
    return x_db.clamp(min=-50, max=80)


# ----------------------------------------------------------------------------------
# a4: TorchScaler("instance", "minmax", dims=(1, 2))  (desed_task/utils/scaler.py:107-120)
# ----------------------------------------------------------------------------------
def scale_minmax(x: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    mn = torch.amin(x, dim=(1, 2), keepdim=True)
    mx = torch.amax(x, dim=(1, 2), keepdim=True)
    return (x - mn) / (mx - mn + eps) * 2 - 1


# ----------------------------------------------------------------------------------
# a2: mixup (desed_task/data_augm.py:19-53) with the random draws (c, perm) injected
# ----------------------------------------------------------------------------------
def mixup_apply(data: torch.Tensor, target: torch.Tensor, c: float, perm: torch.Tensor, label_type="soft"):
    mixed = c * data + (1 - c) * data[perm, :]
    if label_type == "soft":
        tgt = torch.clamp(c * target + (1 - c) * target[perm, :], min=0, max=1)
    elif label_type == "hard":
        tgt = torch.clamp(target + target[perm, :], min=0, max=1)
    else:
        raise NotImplementedError(label_type)
    return mixed, tgt


# ----------------------------------------------------------------------------------
# a6: SpecAugment (desed_task/nnet/CRNN.py:207-219) = two torchaudio axis masks.
# mask_param = min(l, int(axis_len * p)); v = rand*mask_param; s = rand*(axis_len - v);
# zero [floor(s), floor(s)+floor(v)).  The draws are injected so tests are deterministic.
# ----------------------------------------------------------------------------------
def specaug_mask_param(length_cap: int, p: float, axis_len: int) -> int:
    return min(length_cap, int(axis_len * p))


def specaug_bounds(u_value: torch.Tensor, u_start: torch.Tensor, mask_param: int, axis_len: int):
    """u_value,u_start: uniform [0,1) draws (per clip, or 1 element for the shared-mask mode).
    Returns integer [start, end) per clip following torchaudio.functional.mask_along_axis_iid."""
    value = u_value * mask_param
    min_value = u_start * (axis_len - value)
    start = min_value.long()
    end = min_value.long() + value.long()
    return start, end


def specaug_apply(x: torch.Tensor, f_bounds, t_bounds) -> torch.Tensor:
    """x (B, F, T); f_bounds/t_bounds = (start, end) int tensors of shape (B,) (or (1,) shared)."""
    B, Fq, T = x.shape
    fi = torch.arange(Fq).view(1, Fq, 1)
    ti = torch.arange(T).view(1, 1, T)
    fs, fe = [b.view(-1, 1, 1) for b in f_bounds]
    ts, te = [b.view(-1, 1, 1) for b in t_bounds]
    x = x.masked_fill((fi >= fs) & (fi < fe), 0.0)     # frequency mask first (CRNN.py:218)
    x = x.masked_fill((ti >= ts) & (ti < te), 0.0)
    return x


# ----------------------------------------------------------------------------------
# a8-a10: CRNN forward, functional over a reference-layout state dict
# ----------------------------------------------------------------------------------
def time_mask(x: torch.Tensor, bounds) -> torch.Tensor:
    """torchaudio TimeMasking on the frame axis of a (B, T, C) tensor with the draws injected: frames [start, end) of clip b are
    zeroed (`dropstep_recurrent`, CRNN.py:288-301).  bounds = (start, end) int tensors of shape (B,) or None."""
    if bounds is None:
        return x
    ti = torch.arange(x.shape[1]).view(1, -1, 1)
    s, e = [b.view(-1, 1, 1) for b in bounds]
    return x.masked_fill((ti >= s) & (ti < e), 0.0)


def crnn_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, training: bool = False,
                 drop_masks: Optional[Sequence[Optional[torch.Tensor]]] = None, dropout_p: float = 0.5,
                 update_bn: bool = True, taps: Optional[dict] = None, embeddings: Optional[torch.Tensor] = None,
                 classes_mask: Optional[torch.Tensor] = None, pad_mask: Optional[torch.Tensor] = None,
                 dropstep=None, aggregation_type: str = "pool1d"):
    """x: (B, n_mels, T) scaled log-mel (SpecAugment, if any, already applied).
    drop_masks: None -> no dropout (even when training); else a list of 8 keep-masks
    (7 CNN blocks in NHWC-agnostic NCHW shape (B,C,T,F) pre-pool, then the post-GRU (B,T',256)),
    applied as x * mask / (1-p) (inverted dropout, CNN.py:90-91, CRNN.py:103,304).
    In training mode BN uses batch stats and (if update_bn) updates running stats in sd in place.
    embeddings (B, E, Te): the `use_embeddings` branch, aggregation_type "pool1d" (CRNN.py:283-296) or "interpolate" (:271-279),
    with sd["cat_tf.*"]; its dropout mask is drop_masks[8] of shape (B, T', C + E).
    dropstep: `dropstep_recurrent` draws (training only): with embeddings a pair ((sx, ex), (se, ee)) -- the CNN features' span
    and, drawn second, the embeddings' (:292-293); without, one (s, e) pair, and the GRU input is then ALSO dropped out
    (:296-301) with mask drop_masks[8] of shape (B, T', C).
    classes_mask (B, nclass) bool, True = class annotated in the clip's data set; pad_mask (B, 1, T') bool, True = padded
    frame: CRNN.py:157-176.
    Returns strong (B, nclass, T//4), weak (B, nclass).  Follows CRNN.py:221-306, CNN.py:66-98."""
    h = x.transpose(1, 2).unsqueeze(1)                                   # (B,1,T,F)  CRNN.py:224
    for i in range(len(NB_FILTERS)):
        p = f"cnn.cnn."
        h = F.conv2d(h, sd[p + f"conv{i}.weight"], sd[p + f"conv{i}.bias"], stride=1, padding=1)
        rm, rv = sd[p + f"batchnorm{i}.running_mean"], sd[p + f"batchnorm{i}.running_var"]
        if training and not update_bn:
            rm, rv = rm.clone(), rv.clone()
        h = F.batch_norm(h, rm, rv, sd[p + f"batchnorm{i}.weight"], sd[p + f"batchnorm{i}.bias"],
                         training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
        if taps is not None:
            taps[f"bn{i}"] = h
        lin = F.linear(h.permute(0, 2, 3, 1), sd[p + f"glu{i}.linear.weight"], sd[p + f"glu{i}.linear.bias"])
        h = lin.permute(0, 3, 1, 2) * torch.sigmoid(h)                  # GLU, CNN.py:11-16
        if drop_masks is not None and drop_masks[i] is not None:
            h = h * drop_masks[i] / (1.0 - dropout_p)
        h = F.avg_pool2d(h, POOLING[i])
        if taps is not None:
            taps[f"block{i}"] = h
    h = h.squeeze(-1).permute(0, 2, 1)                                   # (B,T',C)  CRNN.py:244-245
    if embeddings is not None:
        if aggregation_type == "interpolate":                                            # CRNN.py:271-279
            reshape_emb = F.interpolate(embeddings.unsqueeze(1), size=(embeddings.shape[1], h.shape[1]),
                                        mode="nearest-exact").squeeze(1).transpose(1, 2)
        else:
            reshape_emb = F.adaptive_avg_pool1d(embeddings, h.shape[1]).transpose(1, 2)  # CRNN.py:283-286
        if dropstep is not None and training:                                            # CRNN.py:288-294
            h = time_mask(h, dropstep[0])
            reshape_emb = time_mask(reshape_emb, dropstep[1])
        z = torch.cat((h, reshape_emb), -1)
        if drop_masks is not None and len(drop_masks) > 8 and drop_masks[8] is not None:
            z = z * drop_masks[8] / (1.0 - dropout_p)
        h = F.linear(z, sd["cat_tf.weight"], sd["cat_tf.bias"])          # CRNN.py:296
        if taps is not None:
            taps["cat_tf"] = h
    elif dropstep is not None and training:                                              # CRNN.py:296-301
        h = time_mask(h, dropstep)
        if drop_masks is not None and len(drop_masks) > 8 and drop_masks[8] is not None:
            h = h * drop_masks[8] / (1.0 - dropout_p)
    flat = []
    for layer in range(2):
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                flat.append(sd[f"rnn.rnn.{nm}_l{layer}{sfx}"])
    H = sd["rnn.rnn.weight_hh_l0"].shape[1]
    h0 = torch.zeros(4, h.shape[0], H, dtype=h.dtype)
    # nn.GRU(batch_first=True, bidirectional=True, num_layers=2), RNN.py:19-30
    h, _ = torch._VF.gru(h, h0, flat, True, 2, 0.0, False, True, True)
    if taps is not None:
        taps["gru"] = h
    if drop_masks is not None and drop_masks[7] is not None:
        h = h * drop_masks[7] / (1.0 - dropout_p)
    strong = torch.sigmoid(F.linear(h, sd["dense.weight"], sd["dense.bias"]))          # CRNN.py:155-156
    invalid = None
    if classes_mask is not None:
        invalid = ~classes_mask[:, None].expand_as(strong)                             # CRNN.py:157-158
    sof = F.linear(h, sd["dense_softmax.weight"], sd["dense_softmax.bias"])
    if pad_mask is not None:
        sof = sof.masked_fill(pad_mask.transpose(1, 2), -1e30)                         # CRNN.py:161-162
    if invalid is not None:
        sof = sof.masked_fill(invalid, -1e30)                                          # CRNN.py:164-166
    sof = torch.softmax(sof, dim=-1)                                                   # over CLASSES (CRNN.py:125)
    sof = torch.clamp(sof, min=1e-7, max=1)
    weak = (strong * sof).sum(1) / sof.sum(1)
    if invalid is not None:                                                            # CRNN.py:173-176
        strong = strong.masked_fill(invalid, 0.0)
        weak = weak.masked_fill(invalid[:, 0], 0.0)
    return strong.transpose(1, 2), weak


def gru_reference_loop(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Explicit single-direction GRU (gate order r,z,n; n = tanh(Wx + b_in + r*(Wh + b_hn))).
    x (B,T,I) -> (B,T,H).  Used to cross-check torch._VF.gru in the tests (SURVEY 8a footnote)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H, dtype=x.dtype)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = F.linear(x[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1)


# ----------------------------------------------------------------------------------
# parameter initialisation with the reference's (torch default) distributions and key layout
# ----------------------------------------------------------------------------------
def crnn_param_shapes(n_in=1, nclass=10, nb_filters=NB_FILTERS, hidden=128, embedding_size=None):
    """embedding_size: adds `cat_tf` (CRNN.py:143-144, aggregation_type "pool1d"), registered after the heads."""
    shapes = {}
    cin = n_in
    for i, co in enumerate(nb_filters):
        shapes[f"cnn.cnn.conv{i}.weight"] = (co, cin, 3, 3)
        shapes[f"cnn.cnn.conv{i}.bias"] = (co,)
        shapes[f"cnn.cnn.batchnorm{i}.weight"] = (co,)
        shapes[f"cnn.cnn.batchnorm{i}.bias"] = (co,)
        shapes[f"cnn.cnn.glu{i}.linear.weight"] = (co, co)
        shapes[f"cnn.cnn.glu{i}.linear.bias"] = (co,)
        cin = co
    for layer in range(2):
        isz = nb_filters[-1] if layer == 0 else 2 * hidden
        for sfx in ("", "_reverse"):
            shapes[f"rnn.rnn.weight_ih_l{layer}{sfx}"] = (3 * hidden, isz)
            shapes[f"rnn.rnn.weight_hh_l{layer}{sfx}"] = (3 * hidden, hidden)
            shapes[f"rnn.rnn.bias_ih_l{layer}{sfx}"] = (3 * hidden,)
            shapes[f"rnn.rnn.bias_hh_l{layer}{sfx}"] = (3 * hidden,)
    shapes["dense.weight"] = (nclass, 2 * hidden)
    shapes["dense.bias"] = (nclass,)
    shapes["dense_softmax.weight"] = (nclass, 2 * hidden)
    shapes["dense_softmax.bias"] = (nclass,)
    if embedding_size is not None:
        shapes["cat_tf.weight"] = (nb_filters[-1], nb_filters[-1] + embedding_size)
        shapes["cat_tf.bias"] = (nb_filters[-1],)
    return shapes


def lcg_fill(shape, seed: int, scale: float = 1.0, offset: float = 0.0) -> torch.Tensor:
    """Deterministic closed-form fill in [offset-scale, offset+scale): a 32-bit LCG so that the
    golden generator, the oracle and the GPU tests agree without shipping arrays."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    state = (idx * np.uint64(2654435761) + np.uint64(seed) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    state = (state * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
    state ^= state >> np.uint64(15)
    state = (state * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    state ^= state >> np.uint64(13)
    u = (state >> np.uint64(8)).astype(np.float64) / float(1 << 24)      # [0,1)
    return torch.from_numpy(((u * 2 - 1) * scale + offset).astype(np.float32)).reshape(shape)


def make_state_dict(seed: int = 7, nclass=10, bn_stats: bool = True, embedding_size=None, hidden=128) -> Dict[str, torch.Tensor]:
    """LCG-filled CRNN state dict with magnitudes like torch's default init (U(+-1/sqrt(fan_in)))."""
    sd = {}
    k = seed * 1000
    for name, shp in crnn_param_shapes(nclass=nclass, embedding_size=embedding_size, hidden=hidden).items():
        k += 1
        if "batchnorm" in name:
            sd[name] = lcg_fill(shp, k, 0.25, 1.0) if name.endswith("weight") else lcg_fill(shp, k, 0.1)
        elif name.startswith("rnn."):
            sd[name] = lcg_fill(shp, k, 1.0 / math.sqrt(hidden))
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else None
            if fan_in is None:   # bias: fan_in of the matching weight
                w = sd[name.replace("bias", "weight")]
                fan_in = int(np.prod(w.shape[1:]))
            sd[name] = lcg_fill(shp, k, 1.0 / math.sqrt(fan_in))
    for i, co in enumerate(NB_FILTERS):
        k += 1
        sd[f"cnn.cnn.batchnorm{i}.running_mean"] = lcg_fill((co,), k, 0.2) if bn_stats else torch.zeros(co)
        k += 1
        sd[f"cnn.cnn.batchnorm{i}.running_var"] = lcg_fill((co,), k, 0.3, 1.0) if bn_stats else torch.ones(co)
        sd[f"cnn.cnn.batchnorm{i}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return sd


def tap_sample(t: torch.Tensor) -> torch.Tensor:
    """Strided sample of an NCHW activation (keeps fixtures small); plus use .sum() for a checksum."""
    _, C, T, Fq = t.shape
    return t[:, ::max(1, C // 8), ::5, ::max(1, Fq // 8)]


PARAM_KEYS = [k for k in crnn_param_shapes().keys()]   # parameters() order of the reference module


def param_keys(sd) -> list:
    """Parameter keys of a state dict in the reference's parameters() order (cat_tf last when present)."""
    return PARAM_KEYS + [k for k in ("cat_tf.weight", "cat_tf.bias") if k in sd]


def synth_audio(batch: int, n_samples: int = 160000, seed: int = 1234) -> torch.Tensor:
    """Closed-form synthetic clips: LCG noise (amplitude 0.1) + two chirps."""
    t = torch.arange(n_samples, dtype=torch.float64) / 16000.0
    out = []
    for b in range(batch):
        noise = lcg_fill((n_samples,), seed + 17 * b, 0.1).double()
        f0, f1 = 200.0 + 90.0 * b, 3000.0 + 140.0 * b
        dur = n_samples / 16000.0
        ph1 = 2 * math.pi * (f0 * t + 0.5 * (f1 - f0) / dur * t * t)
        ph2 = 2 * math.pi * (f1 * t - 0.5 * (f1 - f0) / dur * t * t)
        out.append((noise + 0.3 * torch.sin(ph1) + 0.2 * torch.sin(ph2) * (t > 0.3 * dur)).float())
    return torch.stack(out)


def synth_labels(batch_sizes=(12, 12, 24), nclass=10, n_frames=156, seed=99) -> torch.Tensor:
    """(B, nclass, n_frames): strong rows Bernoulli(0.1)-ish, weak rows only frame 0, unlabelled zero
    (batch contract of desed_task/dataio/datasets.py:208-237,333-338,448-449)."""
    ns, nw, nu = batch_sizes
    B = ns + nw + nu
    lab = torch.zeros(B, nclass, n_frames)
    u = (lcg_fill((ns, nclass, n_frames), seed, 0.5, 0.5))
    lab[:ns] = (u < 0.1).float()
    u = lcg_fill((nw, nclass), seed + 1, 0.5, 0.5)
    lab[ns:ns + nw, :, 0] = (u < 0.2).float()
    return lab


# ----------------------------------------------------------------------------------
# a12: ExponentialWarmup (desed_task/utils/schedulers.py:60-104), no annealing
# ----------------------------------------------------------------------------------
def warmup_factor(step_num: int, rampup_len: int, exponent: float = -5.0) -> float:
    if rampup_len == 0:
        return 1.0
    current = float(np.clip(step_num, 0.0, rampup_len))
    phase = 1.0 - current / rampup_len
    return float(np.exp(exponent * phase * phase))


# ----------------------------------------------------------------------------------
# a13: update_ema (sed_trainer.py:187-199)
# ----------------------------------------------------------------------------------
def ema_update(teacher: Dict[str, torch.Tensor], student: Dict[str, torch.Tensor], alpha: float, global_step: int):
    alpha = min(1 - 1 / (global_step + 1), alpha)
    for k in param_keys(student):
        teacher[k].mul_(alpha).add_(student[k], alpha=1 - alpha)
    return alpha


# ----------------------------------------------------------------------------------
# a14: Adam (torch.optim.Adam defaults, train_sed.py:199-201), plain restatement
# ----------------------------------------------------------------------------------
def adam_step(param, grad, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------
# a16: the whole training step (sed_trainer.py:269-365 + Lightning 1.9 hook order)
# ----------------------------------------------------------------------------------
class OracleTrainer:
    """Unfused CPU mean-teacher trainer.  Random draws (mixup gate/c/perm, SpecAugment, dropout)
    are injected through `rng` dicts so the HIP path can be compared on identical draws."""

    def __init__(self, student_sd, batch_sizes=(12, 12, 24), lr=1e-3, rampup_len=5900, const_max=2.0,
                 ema_factor=0.999, dropout_p=0.5, teacher_sd=None, self_sup="mse"):
        self.student = {k: v.clone() for k, v in student_sd.items()}
        self.keys = param_keys(student_sd)
        for k in self.keys:
            self.student[k].requires_grad_(True)
        src = teacher_sd if teacher_sd is not None else student_sd
        self.teacher = {k: v.clone() for k, v in src.items()}          # deepcopy (sed_trainer.py:61-64)
        self.batch_sizes = batch_sizes
        self.max_lr, self.rampup_len, self.const_max, self.ema_factor = lr, rampup_len, const_max, ema_factor
        self.dropout_p = dropout_p
        self.selfsup_loss = F.mse_loss if self_sup == "mse" else F.binary_cross_entropy      # sed_trainer.py:97-100
        self.step_num = 1                                              # schedulers.py:79
        self.lr = lr
        self.m = {k: torch.zeros_like(self.student[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(self.student[k]) for k in self.keys}
        self.adam_steps = 0

    def features(self, audio, labels, mix=None):
        ns, nw, _ = self.batch_sizes
        feats = mel_spectrogram(audio)
        labels = labels.clone()
        labels_weak = (labels[ns:ns + nw].sum(-1) > 0).float()
        if mix is not None:                                            # weak first, then strong (:294-301)
            feats[ns:ns + nw], labels_weak = mixup_apply(feats[ns:ns + nw], labels_weak, mix["c_weak"], mix["perm_weak"])
            feats[:ns], labels[:ns] = mixup_apply(feats[:ns], labels[:ns], mix["c_strong"], mix["perm_strong"])
        return feats, labels, labels_weak

    def detect(self, feats, sd, training, aug=None, drop_masks=None, update_bn=True, embeddings=None):
        x = scale_minmax(take_log(feats))
        if aug is not None:
            x = specaug_apply(x, aug["f"], aug["t"])
        return crnn_forward(sd, x, training=training, drop_masks=drop_masks, dropout_p=self.dropout_p,
                            update_bn=update_bn, embeddings=embeddings)

    def training_step(self, audio, labels, mix=None, aug_s=None, aug_t=None, drop_s=None, drop_t=None, embeddings=None):
        """embeddings: the pretrained variant of the step (sed_trainer_pretrained.py:282-400) -- the same embeddings go to
        the student and the teacher and are NOT mixed up with the features."""
        ns, nw, _ = self.batch_sizes
        feats, labels, labels_weak = self.features(audio, labels, mix)
        strong_s, weak_s = self.detect(feats, self.student, True, aug_s, drop_s, embeddings=embeddings)
        loss_strong = F.binary_cross_entropy(strong_s[:ns], labels[:ns])
        loss_weak = F.binary_cross_entropy(weak_s[ns:ns + nw], labels_weak)
        with torch.no_grad():
            strong_t, weak_t = self.detect(feats, self.teacher, True, aug_t, drop_t, embeddings=embeddings)   # TRAIN mode (Q7)
            loss_strong_t = F.binary_cross_entropy(strong_t[:ns], labels[:ns])
            loss_weak_t = F.binary_cross_entropy(weak_t[ns:ns + nw], labels_weak)
        weight = self.const_max * warmup_factor(self.step_num, self.rampup_len)
        strong_self = self.selfsup_loss(strong_s, strong_t)
        weak_self = self.selfsup_loss(weak_s, weak_t)
        tot_self = (strong_self + weak_self) * weight
        tot = loss_strong + loss_weak + tot_self
        logs = {
            "train/student/loss_strong": loss_strong.item(), "train/student/loss_weak": loss_weak.item(),
            "train/teacher/loss_strong": loss_strong_t.item(), "train/teacher/loss_weak": loss_weak_t.item(),
            "train/step": self.step_num, "train/student/tot_self_loss": tot_self.item(), "train/weight": weight,
            "train/student/tot_supervised": strong_self.item(),          # sic, Q11
            "train/student/weak_self_sup_loss": weak_self.item(),
            "train/student/strong_self_sup_loss": strong_self.item(), "train/lr": self.lr,
        }
        self.last = dict(strong_s=strong_s.detach(), weak_s=weak_s.detach(), strong_t=strong_t, weak_t=weak_t)
        return tot, logs

    def training_step_2024(self, audio, labels, embeddings, valid, mix=None, const_weight=False):
        """The 2024 recipe's multi-data-set step, recipes/dcase2024_task4_baseline/local/sed_trainer_pretrained.py:318-430.
        batch_sizes = (maestro, synth, strong, weak, unlabelled).  mix: None (mixup gated off) or the six (c, perm) draws of
        apply_mixup in call order -- weak group features, weak group embeddings, synth+strong features, embeddings, maestro
        features, embeddings (:341-351); the labels of a group are mixed by BOTH draws (:283-301).  valid (B, nclass) bool."""
        i_m, i_sy, i_st, i_w, i_u = np.cumsum(self.batch_sizes)
        feats = mel_spectrogram(audio)
        labels, embeddings = labels.clone(), embeddings.clone()
        if mix is not None:
            groups = ((i_st, i_w), (i_m, i_st), (0, i_m))
            for gi, (a, b) in enumerate(groups):
                (c1, p1), (c2, p2) = mix[2 * gi], mix[2 * gi + 1]
                feats[a:b], labels[a:b] = mixup_apply(feats[a:b], labels[a:b], c1, p1)
                embeddings[a:b], labels[a:b] = mixup_apply(embeddings[a:b], labels[a:b], c2, p2)
        labels_weak = (labels[i_st:i_w].sum(-1) > 0).float()
        labels = labels.masked_fill(~valid[:, :, None].expand_as(labels), 0.0)
        labels_weak = labels_weak.masked_fill(~valid[i_st:i_w], 0.0)
        x = scale_minmax(take_log(feats))
        strong_s, weak_s = crnn_forward(self.student, x, training=True, embeddings=embeddings, classes_mask=valid)
        loss_strong = F.binary_cross_entropy(strong_s[:i_st], labels[:i_st])
        loss_weak = F.binary_cross_entropy(weak_s[i_st:i_w], labels_weak)
        with torch.no_grad():
            strong_t, weak_t = crnn_forward(self.teacher, x, training=True, embeddings=embeddings, classes_mask=valid)
        weight = self.const_max * (1.0 if const_weight else warmup_factor(self.step_num, self.rampup_len))      # :393-396
        strong_self = self.selfsup_loss(strong_s[i_m:], strong_t[i_m:])
        weak_self = self.selfsup_loss(weak_s[i_m:], weak_t[i_m:])
        tot_self = (strong_self + weak_self) * weight
        tot = loss_strong + loss_weak + tot_self
        logs = {
            "train/student/loss_strong": loss_strong.item(), "train/student/loss_weak": loss_weak.item(),
            "train/step": self.step_num, "train/student/tot_self_loss": tot_self.item(), "train/weight": weight,
            "train/student/tot_supervised": strong_self.item(), "train/student/weak_self_sup_loss": weak_self.item(),
            "train/student/strong_self_sup_loss": strong_self.item(), "train/lr": self.lr,
        }
        self.last = dict(strong_s=strong_s.detach(), weak_s=weak_s.detach(), strong_t=strong_t, weak_t=weak_t)
        return tot, logs

    def optimizer_step(self, tot_loss):
        """Lightning 1.9 order: on_before_zero_grad(EMA) -> zero_grad -> backward -> Adam -> scheduler."""
        with torch.no_grad():
            ema_update(self.teacher, {k: self.student[k].detach() for k in self.keys}, self.ema_factor, self.step_num)
        params = [self.student[k] for k in self.keys]
        grads = torch.autograd.grad(tot_loss, params, allow_unused=True)
        self.adam_steps += 1
        with torch.no_grad():
            for k, g in zip(self.keys, grads):
                if g is None:
                    g = torch.zeros_like(self.student[k])
                adam_step(self.student[k], g, self.m[k], self.v[k], self.adam_steps, self.lr)
        self.step_num += 1
        self.lr = self.max_lr * warmup_factor(self.step_num, self.rampup_len)
        return dict(zip(self.keys, grads))


# ------------------------------------------------------------------------------------------------
# inference post-processing: recipes/dcase2023_task4_baseline/local/utils.py:16-73
# ------------------------------------------------------------------------------------------------
def median_filter_scores(scores_tc: np.ndarray, win: int = 7) -> np.ndarray:
    """utils.py:55: scipy.ndimage.median_filter(c_scores, (median_filter, 1)) on a (T, NC) array (scipy is the reference's
    own dependency here and is importable in this image: this is the reference call, not a restatement)."""
    import scipy.ndimage
    return scipy.ndimage.median_filter(scores_tc, (win, 1))


def find_contiguous_regions(activity: np.ndarray) -> np.ndarray:
    """dcase_util.data.DecisionEncoder.find_contiguous_regions (third party, pinned by the reference only as `dcase_util`
    in requirements; called at desed_task/utils/encoder.py:200).  Published algorithm, restated: parity unpinned."""
    activity = np.asarray(activity).astype(bool)
    change = np.logical_xor(activity[1:], activity[:-1]).nonzero()[0] + 1
    if activity.size and activity[0]:
        change = np.r_[0, change]
    if activity.size and activity[-1]:
        change = np.r_[change, activity.size]
    return change.reshape((-1, 2))


def decode_events(scores_tc: np.ndarray, threshold: float):
    """utils.py:62-63 + encoder.py:189-211: list of (class index, onset frame, offset frame), class-major."""
    out = []
    pred = scores_tc > threshold
    for c, col in enumerate(pred.T):
        for on, off in find_contiguous_regions(col):
            out.append((c, int(on), int(off)))
    return out
