"""CPU oracle for the frozen BEATs feature extractor.  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU, fp32, unfused) restatement of the inference path of
recipes/dcase2023_task4_baseline/local/beats/BEATs.py:135-204 (`BEATs.extract_features`, no padding mask, predictor-less) and
backbone.py:23-160, 214-296, 446-700 (`TransformerEncoder`, post-LayerNorm / deep-norm encoder layer, multi-head attention with
the bucketed relative position bias and its GRU-style gate).  Only tests/ and __graft_entry__ import it.

Pinning: `beats_forward` is pinned on the reference module itself with closed-form weights -- tests/golden/make_golden_beats.py
imports the reference's `BEATs` class, fills it from `make_beats_state_dict` and records its outputs (tests/golden/golden_beats.npz,
checked by tests/test_oracle_golden.py).  `kaldi_fbank` is NOT pinned by the reference: the arithmetic lives in
torchaudio.compliance.kaldi (BEATs.py:15,121-128), which the reference neither vendors nor pins and which is not installed here.  It
follows torchaudio's published algorithm (Kaldi's compute-fbank-feats with these options) and is cross-checked against an
independent float64 numpy implementation in the tests; **parity unpinned** for the filterbank front-end.  No BEATs checkpoint is
available offline (extract_embeddings.py:181-185 downloads it), so parity with trained weights is structural.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .sed_oracle import lcg_fill

# cfg of the checkpoint the recipe downloads (BEATs_iter3_plus_AS2M.pt); only the keys the inference path reads
BEATS_ITER3_CFG = dict(input_patch_size=16, embed_dim=512, conv_bias=False, encoder_layers=12, encoder_embed_dim=768,
                       encoder_ffn_embed_dim=3072, encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False,
                       deep_norm=True, conv_pos=128, conv_pos_groups=16, relative_position_embedding=True, num_buckets=320,
                       max_distance=800, gru_rel_pos=True, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                       encoder_layerdrop=0.0, dropout_input=0.0, finetuned_model=False, layer_wise_gradient_decay_ratio=1.0)
FBANK_MEAN, FBANK_STD = 15.41663, 6.55582          # BEATs.py:112-113


# ----------------------------------------------------------------------------------
# torchaudio.compliance.kaldi.fbank(waveform * 2**15, num_mel_bins=128, sample_frequency=16000, frame_length=25, frame_shift=10)
# (BEATs.py:119-128) with that function's defaults: dither 0, preemphasis 0.97, remove_dc_offset, povey window, snip_edges,
# round_to_power_of_two (512-point FFT), low_freq 20, high_freq 0 (= Nyquist), use_power, use_log_fbank, no energy column.
# ----------------------------------------------------------------------------------
def kaldi_mel_banks(num_bins=128, n_fft=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0) -> torch.Tensor:
    """(num_bins, n_fft // 2) triangular filters on Kaldi's mel scale 1127 ln(1 + f / 700)."""
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)      # noqa: E731
    nyq = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyq
    n_bins_fft = n_fft // 2
    bin_width = sample_freq / n_fft
    mel_lo, mel_hi = mel(low_freq), mel(high_freq)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1.0) * delta, mel_lo + (b + 2.0) * delta
    melf = 1127.0 * torch.log(1.0 + bin_width * torch.arange(n_bins_fft, dtype=torch.float32) / 700.0).unsqueeze(0)
    up = (melf - left) / (center - left)
    down = (right - melf) / (right - center)
    return torch.clamp(torch.min(up, down), min=0.0)


def kaldi_fbank(waveform: torch.Tensor, num_mel_bins=128, frame_length=400, frame_shift=160, n_fft=512, preemph=0.97) -> torch.Tensor:
    """waveform (N,) already scaled by 2**15 -> (1 + (N - 400) // 160, 128) log mel energies."""
    n = waveform.shape[0]
    m = 1 + (n - frame_length) // frame_shift
    frames = waveform.unfold(0, frame_length, frame_shift)[:m].clone()                  # snip_edges
    frames = frames - frames.mean(dim=1, keepdim=True)                                   # remove_dc_offset
    prev = torch.cat((frames[:, :1], frames[:, :-1]), 1)                                 # replicate-padded shift
    frames = frames - preemph * prev
    window = torch.hann_window(frame_length, periodic=False, dtype=torch.float32).pow(0.85)      # "povey"
    frames = F.pad(frames * window, (0, n_fft - frame_length))
    power = torch.fft.rfft(frames).abs().pow(2.0)                                        # (m, 257)
    banks = F.pad(kaldi_mel_banks(num_mel_bins, n_fft), (0, 1))                          # zero column for the Nyquist bin
    mel = power @ banks.T
    return torch.clamp(mel, min=torch.finfo(torch.float32).eps).log()


def beats_preprocess(source: torch.Tensor) -> torch.Tensor:
    """BEATs.preprocess (BEATs.py:109-133): (B, N) waveforms -> normalised fbank (B, frames, 128)."""
    fb = torch.stack([kaldi_fbank(w * 2 ** 15) for w in source])
    return (fb - FBANK_MEAN) / (2 * FBANK_STD)


# ----------------------------------------------------------------------------------
# relative position bias (backbone.py:390-444) and its gate (:662-682)
# ----------------------------------------------------------------------------------
def relative_buckets(q_len: int, k_len: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    rel = torch.arange(k_len)[None, :] - torch.arange(q_len)[:, None]
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


def beats_forward(sd: Dict[str, torch.Tensor], cfg: dict, fbank: torch.Tensor, taps=None) -> torch.Tensor:
    """fbank (B, frames, 128) normalised -> features (B, tokens, encoder_embed_dim).  Eval mode (no dropout / layerdrop)."""
    P, D, H = cfg["input_patch_size"], cfg["encoder_embed_dim"], cfg["encoder_attention_heads"]
    x = F.conv2d(fbank.unsqueeze(1), sd["patch_embedding.weight"], sd.get("patch_embedding.bias"), stride=P)    # BEATs.py:153-154
    x = x.reshape(x.shape[0], x.shape[1], -1).transpose(1, 2)                                                   # token = t * 8 + f
    x = F.layer_norm(x, (cfg["embed_dim"],), sd["layer_norm.weight"], sd["layer_norm.bias"])
    if "post_extract_proj.weight" in sd:
        x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    if taps is not None:
        taps["proj"] = x
    # positional convolution (backbone.py:30-43,118-120): weight-normalised grouped Conv1d + SamePad + GELU, added to x
    g, v = sd["encoder.pos_conv.0.weight_g"], sd["encoder.pos_conv.0.weight_v"]
    w = v * (g / v.norm(dim=(0, 1), keepdim=True))                          # weight_norm(dim=2): one norm per kernel tap
    K = cfg["conv_pos"]
    pc = F.conv1d(x.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=K // 2, groups=cfg["conv_pos_groups"])
    if K % 2 == 0:
        pc = pc[:, :, :-1]
    x = x + F.gelu(pc).transpose(1, 2)
    if not cfg["layer_norm_first"]:
        x = F.layer_norm(x, (D,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"])
    if taps is not None:
        taps["enc_in"] = x
    B, T, _ = x.shape
    hd = D // H
    alpha_dn = math.pow(2 * cfg["encoder_layers"], 0.25) if cfg["deep_norm"] else 1.0
    pos_bias = None
    if cfg["relative_position_embedding"]:
        buckets = relative_buckets(T, T, cfg["num_buckets"], cfg["max_distance"])
        pos_bias = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"][buckets].permute(2, 0, 1)      # (H, T, T)
    for i in range(cfg["encoder_layers"]):
        p = "encoder.layers.%d." % i
        res = x
        if cfg["layer_norm_first"]:
            raise NotImplementedError("the BEATs checkpoints of the recipe are post-LayerNorm (layer_norm_first: False)")
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v_ = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        qh = q.view(B, T, H, hd).transpose(1, 2)                            # (B, H, T, hd), unscaled
        kh = k.view(B, T, H, hd).transpose(1, 2)
        vh = v_.view(B, T, H, hd).transpose(1, 2)
        # backbone.py:529-531, :640-643: q * scaling / 32, (scores - rowmax) * 32  ==  q.k * scaling up to the softmax's shift
        scores = (qh * (hd ** -0.5)) @ kh.transpose(-1, -2)
        if pos_bias is not None:
            bias = pos_bias.unsqueeze(0)
            if cfg["gru_rel_pos"]:
                gl = F.linear(qh, sd[p + "self_attn.grep_linear.weight"], sd[p + "self_attn.grep_linear.bias"])
                ga, gb = torch.sigmoid(gl.view(B, H, T, 2, 4).sum(-1)).chunk(2, dim=-1)
                gate = ga * (gb * sd[p + "self_attn.grep_a"] - 1.0) + 2.0                  # (B, H, T, 1)
                bias = gate * bias
            scores = scores + bias
        attn = torch.softmax(scores, dim=-1) @ vh                                           # (B, H, T, hd)
        attn = attn.transpose(1, 2).reshape(B, T, D)
        attn = F.linear(attn, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(res * alpha_dn + attn, (D,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"])
        res = x
        h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        h = F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        x = F.layer_norm(res * alpha_dn + h, (D,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"])
        if taps is not None:
            taps["layer%d" % i] = x
    return x


def beats_embeddings(sd, cfg, source: torch.Tensor):
    """BEATsModel.forward (BEATs.py:216-223): {"global": (B, D), "frame": (B, D, tokens)}."""
    feats = beats_forward(sd, cfg, beats_preprocess(source))
    return {"global": feats.mean(dim=1), "frame": feats.transpose(1, 2)}


# ----------------------------------------------------------------------------------
# closed-form weights in the reference module's state-dict layout
# ----------------------------------------------------------------------------------
def beats_param_shapes(cfg: dict) -> Dict[str, tuple]:
    P, E, D, Fd, H = cfg["input_patch_size"], cfg["embed_dim"], cfg["encoder_embed_dim"], cfg["encoder_ffn_embed_dim"], cfg["encoder_attention_heads"]
    s = {}
    if E != D:
        s["post_extract_proj.weight"] = (D, E); s["post_extract_proj.bias"] = (D,)
    s["patch_embedding.weight"] = (E, 1, P, P)
    if cfg["conv_bias"]:
        s["patch_embedding.bias"] = (E,)
    s["encoder.pos_conv.0.bias"] = (D,)
    s["encoder.pos_conv.0.weight_g"] = (1, 1, cfg["conv_pos"])
    s["encoder.pos_conv.0.weight_v"] = (D, D // cfg["conv_pos_groups"], cfg["conv_pos"])
    for i in range(cfg["encoder_layers"]):
        p = "encoder.layers.%d." % i
        if cfg["relative_position_embedding"]:      # ONE embedding shared by all layers (backbone.py:78-83): same tensor under every key
            s[p + "self_attn.relative_attention_bias.weight"] = (cfg["num_buckets"], H)
        if cfg["gru_rel_pos"]:
            s[p + "self_attn.grep_a"] = (1, H, 1, 1)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + "self_attn.%s.weight" % n] = (D, D); s[p + "self_attn.%s.bias" % n] = (D,)
        if cfg["gru_rel_pos"]:
            s[p + "self_attn.grep_linear.weight"] = (8, D // H); s[p + "self_attn.grep_linear.bias"] = (8,)
        s[p + "self_attn_layer_norm.weight"] = (D,); s[p + "self_attn_layer_norm.bias"] = (D,)
        s[p + "fc1.weight"] = (Fd, D); s[p + "fc1.bias"] = (Fd,)
        s[p + "fc2.weight"] = (D, Fd); s[p + "fc2.bias"] = (D,)
        s[p + "final_layer_norm.weight"] = (D,); s[p + "final_layer_norm.bias"] = (D,)
    s["encoder.layer_norm.weight"] = (D,); s["encoder.layer_norm.bias"] = (D,)
    s["layer_norm.weight"] = (E,); s["layer_norm.bias"] = (E,)
    return s


def make_beats_state_dict(cfg: dict, seed: int = 3) -> Dict[str, torch.Tensor]:
    """LCG-filled BEATs weights with trained-model-like magnitudes (so that attention is neither uniform nor one-hot)."""
    sd = {}
    k = seed * 100000
    for name, shp in beats_param_shapes(cfg).items():
        k += 1
        if name.endswith("layer_norm.weight") or name.endswith("_layer_norm.weight"):
            sd[name] = lcg_fill(shp, k, 0.2, 1.0)
        elif name.endswith("weight_g"):
            sd[name] = lcg_fill(shp, k, 0.3, 1.5)
        elif name.endswith("grep_a"):
            sd[name] = lcg_fill(shp, k, 0.3, 1.0)
        elif name.endswith("relative_attention_bias.weight"):
            first = "encoder.layers.0.self_attn.relative_attention_bias.weight"
            sd[name] = sd[first] if name != first else lcg_fill(shp, k, 1.5)
        elif name.endswith(".bias"):
            sd[name] = lcg_fill(shp, k, 0.1)
        else:
            fan_in = int(np.prod(shp[1:]))
            sd[name] = lcg_fill(shp, k, 1.7 / math.sqrt(fan_in))
    return sd
