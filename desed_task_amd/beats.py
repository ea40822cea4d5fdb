"""Frozen BEATs feature extractor on the MI355X kernels -- drop-in for the inference surface of
recipes/dcase2023_task4_baseline/local/beats/BEATs.py: `BEATsConfig`, `BEATs` (same state-dict keys, `preprocess`,
`extract_features`) and `BEATsModel(cfg_path)` whose `forward(x)` returns `{"global": (B, D), "frame": (B, D, tokens)}`
(BEATs.py:205-223), i.e. the embeddings `extract_embeddings.py:46-51` writes and the embedding-fusion CRNN consumes.

What runs where (csrc/sed_beats.hip, csrc/sed_gemm_bf16.hip): Kaldi fbank -> 16 x 16 patch gather -> patch embedding, every
Linear and the FFN (GELU in the GEMM epilogue) on the split-bf16 MFMA -> LayerNorm / deep-norm residual kernel -> grouped
position convolution -> fused attention with the gated relative position bias.  Inference only (the recipes keep the extractor
frozen: `pretrained.freezed: True`); the modules below HOLD the parameters under the reference's names so that the published
checkpoints load unchanged, they never run torch arithmetic.

Not built: padding masks, `layer_norm_first` checkpoints, the fine-tuned predictor head, training of the extractor
(`pretrained.e2e`) -- they raise.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib


POSCONV_MFMA = True             # position convolution on the split-bf16 MFMA (False: the f32 vector-pipe kernel, A/B and tests)
LINEAR_TILES = True             # round 6: the q / k / v projection through the LDS-DMA kernel on K-tiled bf16 plane images (sed_linear_tiles_bf16x3); the
                                # LayerNorm in front of it writes the activation's image beside its fp32 output (False: the round-5 path, A/B and tests)
LINEAR_TILES_FFN = True         # ... and fc1 -> GELU -> fc2 the same way (fc1 writes its output as fc2's image, no fp32 copy of the 3072-wide hidden state)
LINEAR_TILES_KSPLIT = True      # ... fc2 (N = 768) as two K halves whose partial sums the final LayerNorm adds (round quantisation: 2 rounds -> 1.5)
LINEAR_PACKED = True            # the large Linear layers through the packed-weight 256 x 128-tile kernel (False: sed_linear_bf16x3, A/B and tests)


class BEATsConfig:
    """Same fields and defaults as the reference's BEATsConfig (BEATs.py:24-86); `update(cfg)` takes the checkpoint's dict."""

    def __init__(self, cfg=None):
        self.input_patch_size = -1
        self.embed_dim = 512
        self.conv_bias = False
        self.encoder_layers = 12
        self.encoder_embed_dim = 768
        self.encoder_ffn_embed_dim = 3072
        self.encoder_attention_heads = 12
        self.activation_fn = "gelu"
        self.layer_wise_gradient_decay_ratio = 1.0
        self.layer_norm_first = False
        self.deep_norm = False
        self.dropout = 0.1
        self.attention_dropout = 0.1
        self.activation_dropout = 0.0
        self.encoder_layerdrop = 0.0
        self.dropout_input = 0.0
        self.conv_pos = 128
        self.conv_pos_groups = 16
        self.relative_position_embedding = False
        self.num_buckets = 320
        self.max_distance = 1280
        self.gru_rel_pos = False
        self.finetuned_model = False
        self.predictor_dropout = 0.1
        self.predictor_class = 527
        if cfg is not None:
            self.update(cfg)

    def update(self, cfg):
        self.__dict__.update(cfg)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the arithmetic runs in the HIP kernels (BEATs.extract_features)")


class _WeightNormConv(_Holder):
    """`nn.utils.weight_norm(nn.Conv1d(D, D, K, groups), dim=2)` as parameters weight_g (1,1,K), weight_v (D, D/groups, K), bias."""

    def __init__(self, D, K, groups):
        super().__init__()
        std = math.sqrt(4.0 / (K * D))
        v = torch.randn(D, D // groups, K) * std
        self.bias = nn.Parameter(torch.zeros(D))
        self.weight_g = nn.Parameter(v.norm(dim=(0, 1), keepdim=True).clone())
        self.weight_v = nn.Parameter(v)


class _Attention(_Holder):
    def __init__(self, D, H, gru_rel_pos, rel_bias):
        super().__init__()
        if rel_bias is not None:
            self.relative_attention_bias = rel_bias                 # ONE embedding shared by all layers (backbone.py:78-83)
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D)
        if gru_rel_pos:
            self.grep_linear = nn.Linear(D // H, 8)
            self.grep_a = nn.Parameter(torch.ones(1, H, 1, 1))


class _Layer(_Holder):
    def __init__(self, cfg, rel_bias):
        super().__init__()
        D = cfg.encoder_embed_dim
        self.self_attn = _Attention(D, cfg.encoder_attention_heads, cfg.gru_rel_pos, rel_bias)
        self.self_attn_layer_norm = nn.LayerNorm(D)
        self.fc1 = nn.Linear(D, cfg.encoder_ffn_embed_dim)
        self.fc2 = nn.Linear(cfg.encoder_ffn_embed_dim, D)
        self.final_layer_norm = nn.LayerNorm(D)


class _Encoder(_Holder):
    def __init__(self, cfg):
        super().__init__()
        D = cfg.encoder_embed_dim
        self.pos_conv = nn.Sequential(_WeightNormConv(D, cfg.conv_pos, cfg.conv_pos_groups))
        rel = nn.Embedding(cfg.num_buckets, cfg.encoder_attention_heads) if cfg.relative_position_embedding else None
        self.layers = nn.ModuleList([_Layer(cfg, rel) for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(D)


def kaldi_mel_banks(num_bins=128, n_fft=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks (no VTLN), float32 like torchaudio: (num_bins, n_fft / 2)."""
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)      # noqa: E731
    if high_freq <= 0.0:
        high_freq += 0.5 * sample_freq
    mel_lo, mel_hi = mel(low_freq), mel(high_freq)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1.0) * delta, mel_lo + (b + 2.0) * delta
    melf = 1127.0 * torch.log(1.0 + (sample_freq / n_fft) * torch.arange(n_fft // 2, dtype=torch.float32) / 700.0).unsqueeze(0)
    return torch.clamp(torch.min((melf - left) / (center - left), (right - melf) / (right - center)), min=0.0)


class KaldiFbank(nn.Module):
    """`ta_kaldi.fbank(waveform * 2**15, num_mel_bins=128, sample_frequency=16000, frame_length=25, frame_shift=10)` for a batch
    (BEATs.py:119-130) + the (x - mean) / (2 std) of BEATs.preprocess, one kernel."""

    def __init__(self, n_mels=128):
        super().__init__()
        self.n_mels = n_mels
        window = torch.hann_window(400, periodic=False, dtype=torch.float32).pow(0.85)
        k = np.arange(256, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / 512), -np.sin(2 * np.pi * k / 512)], 1).astype(np.float32)
        banks = kaldi_mel_banks(n_mels)                                       # (n_mels, 256); the Nyquist bin has weight 0
        start, length = torch.zeros(n_mels, dtype=torch.int32), torch.zeros(n_mels, dtype=torch.int32)
        for m in range(n_mels):
            idx = torch.nonzero(banks[m]).flatten()
            if idx.numel():
                start[m], length[m] = int(idx[0]), int(idx[-1]) - int(idx[0]) + 1
        stride = max(8, int(length.max()))
        w = torch.zeros(n_mels, stride)
        for m in range(n_mels):
            w[m, :int(length[m])] = banks[m, int(start[m]):int(start[m]) + int(length[m])]
        self.fb_stride = stride
        for name, t in (("window", window), ("tw", torch.from_numpy(tw).contiguous()), ("fb_start", start), ("fb_len", length),
                        ("fb_w", w.contiguous())):
            self.register_buffer(name, t, persistent=False)

    def forward(self, source, mean=0.0, std=0.5):
        source = source.float().contiguous()
        _lib.check_tensor(source, "waveforms")
        if self.window.device != source.device:
            self.to(source.device)
        B, N = source.shape
        M = 1 + (N - 400) // 160 if N >= 400 else 0
        out = torch.empty(B, M, self.n_mels, device=source.device, dtype=torch.float32)
        _lib.get().call("sed_kaldi_fbank", source.data_ptr(), out.data_ptr(), B, N, self.n_mels, self.window.data_ptr(),
                        self.tw.data_ptr(), self.fb_start.data_ptr(), self.fb_len.data_ptr(), self.fb_w.data_ptr(), self.fb_stride,
                        float(mean), float(1.0 / (2.0 * std)), _lib.stream_ptr(source))
        return out


class BEATs(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.layer_norm_first:
            raise NotImplementedError("layer_norm_first checkpoints are not built (the recipe's BEATs checkpoints are post-LayerNorm)")
        if cfg.finetuned_model:
            raise NotImplementedError("the fine-tuned predictor head is not built (the recipes use the encoder features)")
        if cfg.activation_fn != "gelu":
            raise NotImplementedError("activation_fn %r: the fused FFN epilogue is GELU" % cfg.activation_fn)
        if cfg.encoder_embed_dim // cfg.encoder_attention_heads != 64:
            raise NotImplementedError("the attention kernel is built for 64-dimensional heads")
        self.cfg = cfg
        self.embed = cfg.embed_dim
        self.post_extract_proj = nn.Linear(self.embed, cfg.encoder_embed_dim) if self.embed != cfg.encoder_embed_dim else None
        self.input_patch_size = cfg.input_patch_size
        self.patch_embedding = nn.Conv2d(1, self.embed, kernel_size=self.input_patch_size, stride=self.input_patch_size, bias=cfg.conv_bias)
        self.encoder = _Encoder(cfg)
        self.layer_norm = nn.LayerNorm(self.embed)
        self.predictor = None
        self.fbank = KaldiFbank(128)
        self._packed = None
        self._relb = {}

    # ---- frozen-weight preparation (once per load) -----------------------------------------------------------------------
    def load_state_dict(self, *a, **k):
        self._packed, self._relb = None, {}
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed, self._relb = None, {}
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        # a PARENT's load_state_dict (BEATsModel, a LightningModule) reaches this module through its recursive `load`, not through
        # load_state_dict above: the derived weight copies (fused q / k / v, position-convolution planes, packed images) are stale then
        self._packed, self._relb = None, {}
        return super()._load_from_state_dict(*a, **k)

    @torch.no_grad()
    def _pack(self):
        if self._packed is not None:
            return self._packed
        cfg, enc = self.cfg, self.encoder
        pc = enc.pos_conv[0]
        v = pc.weight_v.detach().float()
        w = v * (pc.weight_g.detach().float() / v.norm(dim=(0, 1), keepdim=True))                 # weight_norm(dim=2)
        G, CG, K = cfg.conv_pos_groups, v.shape[1], v.shape[2]
        wt = w.view(G, CG, CG, K).permute(0, 3, 1, 2).contiguous()                                # (groups, K, co, ci)
        layers = []
        for lyr in enc.layers:
            a = lyr.self_attn
            layers.append(dict(
                wqkv=torch.cat((a.q_proj.weight, a.k_proj.weight, a.v_proj.weight), 0).detach().float().contiguous(),
                bqkv=torch.cat((a.q_proj.bias, a.k_proj.bias, a.v_proj.bias), 0).detach().float().contiguous(),
                grep_a=a.grep_a.detach().float().reshape(-1).contiguous() if cfg.gru_rel_pos else None))
        # the position convolution's B operand for the split-bf16 MFMA: hi = bf16(w), lo = bf16(w - hi), as bit patterns
        w_hi = wt.to(torch.bfloat16)
        w_lo = (wt - w_hi.float()).to(torch.bfloat16)
        wsplit = torch.stack((w_hi, w_lo)).contiguous().view(torch.int16)
        self._packed = dict(wt=wt, wsplit=wsplit, layers=layers,
                            wpatch=self.patch_embedding.weight.detach().float().reshape(self.embed, -1).contiguous())
        return self._packed

    def _rel_bias(self, T, device):
        """(H, 2T - 1): relative_attention_bias[bucket(s - t)] per offset (backbone.py:390-444)."""
        if not self.cfg.relative_position_embedding:
            return None
        if T not in self._relb:
            cfg = self.cfg
            rel = torch.arange(-(T - 1), T)
            nb = cfg.num_buckets // 2
            out = (rel > 0).long() * nb
            r = rel.abs()
            max_exact = nb // 2
            large = max_exact + (torch.log(r.float() / max_exact) / math.log(cfg.max_distance / max_exact) * (nb - max_exact)).long()
            large = torch.min(large, torch.full_like(large, nb - 1))
            buckets = (out + torch.where(r < max_exact, r, large)).to(device)
            emb = self.encoder.layers[0].self_attn.relative_attention_bias.weight.detach().float()
            self._relb[T] = emb[buckets].t().contiguous()                                          # (H, 2T - 1)
        return self._relb[T]

    # ---- reference surface ---------------------------------------------------------------------------------------------
    def preprocess(self, source, fbank_mean=15.41663, fbank_std=6.55582):
        return self.fbank(source, fbank_mean, fbank_std)

    @torch.no_grad()
    def extract_features(self, source, padding_mask=None, fbank_mean=15.41663, fbank_std=6.55582, taps=None):
        """taps: optional dict that receives the encoder input and every layer's output (tests / diagnostics)."""
        if padding_mask is not None:
            raise NotImplementedError("padding masks are not built (the recipes extract embeddings of fixed 10 s clips)")
        if self.training and (self.cfg.dropout > 0 or self.cfg.encoder_layerdrop > 0 or self.cfg.attention_dropout > 0):
            raise NotImplementedError("the BEATs extractor is an inference path: call .eval() (pretrained.freezed)")
        lib = _lib.get()
        cfg = self.cfg
        pk = self._pack()
        fb = self.preprocess(source, fbank_mean, fbank_std)                       # (B, M, 128)
        st = _lib.stream_ptr(fb)
        B, M, F = fb.shape
        P, E, D, H = cfg.input_patch_size, cfg.embed_dim, cfg.encoder_embed_dim, cfg.encoder_attention_heads
        T = (M // P) * (F // P)
        R = B * T
        f32 = dict(device=fb.device, dtype=torch.float32)

        packed = pk.setdefault("linear", {})

        def linear(x, w, b, n, k, act=0):
            y = torch.empty(x.shape[0], n, **f32)
            if LINEAR_PACKED and n >= 2048 and n % 128 == 0 and k % 32 == 0 and x.shape[0] >= 256:
                # frozen weight: split into bf16 hi / lo planes once, then the 256 x 128-tile kernel (sed_gemm_bf16.hip, round 5).  Wide
                # outputs only: at N = 768 its 558 tiles on 512 resident workgroups lose to the 128-row tiles (gpurun_out/linear_r05a.txt)
                key = (w.data_ptr(), w._version, n, k)      # _version: an in-place weight load keeps the pointer
                wp = packed.get(key)
                if wp is None or wp.device != x.device:
                    wp = torch.empty(2 * n * k, device=x.device, dtype=torch.int16)
                    wsrc = w.detach().float().contiguous()
                    lib.call("sed_pack_weights_bf16x3", wsrc.data_ptr(), wp.data_ptr(), n, k, st)
                    packed[key] = wp
                lib.call("sed_linear_packed_bf16x3", x.data_ptr(), wp.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                         x.shape[0], n, k, act, st)
                return y
            lib.call("sed_linear_bf16x3", x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                     x.shape[0], n, k, act, st)
            return y

        def layernorm(x, res, alpha, ln, d, image=None, x2=None):
            y = torch.empty_like(x)
            if image is not None:
                lib.call("sed_layernorm_tiles", x.data_ptr(), x2.data_ptr() if x2 is not None else None, res.data_ptr() if res is not None else None, float(alpha), ln.weight.data_ptr(),
                         ln.bias.data_ptr(), y.data_ptr(), image.data_ptr(), x.shape[0], d, float(ln.eps), st)
                return y
            lib.call("sed_layernorm", x.data_ptr(), res.data_ptr() if res is not None else None, float(alpha), ln.weight.data_ptr(),
                     ln.bias.data_ptr(), y.data_ptr(), x.shape[0], d, float(ln.eps), st)
            return y

        # round 6: the q / k / v projection reads its activation as a K-tiled bf16 hi / lo image (written by the LayerNorm that produces
        # it) and its frozen weight as the same kind of image (built once); ONE image buffer, re-written by every layer's last LayerNorm
        tiles = LINEAR_TILES and D % 256 == 0 and R >= 256
        # (zeros: the LayerNorm writes rows < R only; the rows of the padded last panel then stay finite all the way through fc1 -> fc2)
        imgs = self._relb.get(("images", R, D, fb.device))                      # (kept across calls: 2 x 73 MB at 48 clips; dropped with the other caches)
        if tiles and imgs is None:
            imgs = self._relb[("images", R, D, fb.device)] = [torch.zeros(2 * ((R + 255) // 256) * 256 * D, device=fb.device, dtype=torch.int16) for _ in range(2)]
        ximg = imgs[0] if tiles else None

        Fd = cfg.encoder_ffn_embed_dim
        ffn_tiles = tiles and LINEAR_TILES_FFN and Fd % 256 == 0
        ximg2 = imgs[1] if ffn_tiles else None                      # the attention block's LayerNorm output -> fc1
        himg = torch.empty(2 * ((R + 255) // 256) * 256 * Fd, device=fb.device, dtype=torch.int16) if ffn_tiles else None   # GELU(fc1) -> fc2

        def linear_tiles(img, w, b, n, k, act=0, out_image=None, split2=False):
            key = ("tiles", w.data_ptr(), w._version, n, k)
            wt_ = packed.get(key)
            if wt_ is None or wt_.device != fb.device:
                wt_ = torch.empty(2 * ((n + 255) // 256) * 256 * k, device=fb.device, dtype=torch.int16)
                wsrc = w.detach().float().contiguous()
                lib.call("sed_split_tiles_bf16x3", wsrc.data_ptr(), wt_.data_ptr(), n, k, st)
                packed[key] = wt_
            bp = b.data_ptr() if b is not None else None
            if out_image is not None:           # the product leaves as the next Linear's image (fc1's GELU output is only read by fc2)
                lib.call("sed_linear_tiles_out_bf16x3", img.data_ptr(), wt_.data_ptr(), bp, out_image.data_ptr(), R, n, k, act, st)
                return None
            if split2:                           # two partial sums over the halves of K: (2, R, n)
                y = torch.empty(2, R, n, **f32)
                lib.call("sed_linear_tiles_split2_bf16x3", img.data_ptr(), wt_.data_ptr(), bp, y.data_ptr(), R, n, k, st)
                return y
            y = torch.empty(R, n, **f32)
            lib.call("sed_linear_tiles_bf16x3", img.data_ptr(), wt_.data_ptr(), bp, y.data_ptr(), R, n, k, act, st)
            return y

        patches = torch.empty(R, P * P, **f32)
        lib.call("sed_patchify", fb.data_ptr(), patches.data_ptr(), B, M, F, P, st)
        x = linear(patches, pk["wpatch"], self.patch_embedding.bias, E, P * P)
        x = layernorm(x, None, 1.0, self.layer_norm, E)
        if self.post_extract_proj is not None:
            x = linear(x, self.post_extract_proj.weight, self.post_extract_proj.bias, D, E)
        enc = self.encoder
        y = torch.empty_like(x)
        if POSCONV_MFMA and D // cfg.conv_pos_groups == 48 and cfg.conv_pos % 4 == 0:
            lib.call("sed_posconv_bf16x3", x.data_ptr(), pk["wsplit"].data_ptr(), enc.pos_conv[0].bias.data_ptr(), y.data_ptr(), B, T, D,
                     cfg.conv_pos, cfg.conv_pos_groups, st)
        else:
            lib.call("sed_posconv", x.data_ptr(), pk["wt"].data_ptr(), enc.pos_conv[0].bias.data_ptr(), y.data_ptr(), B, T, D, cfg.conv_pos,
                     cfg.conv_pos_groups, st)
        x = layernorm(y, None, 1.0, enc.layer_norm, D, image=ximg)
        if taps is not None:
            taps["enc_in"] = x.view(B, T, D)
        alpha = math.pow(2 * cfg.encoder_layers, 0.25) if cfg.deep_norm else 1.0
        relb = self._rel_bias(T, fb.device)
        for lyr, lp in zip(enc.layers, pk["layers"]):
            a = lyr.self_attn
            qkv = linear_tiles(ximg, lp["wqkv"], lp["bqkv"], 3 * D, D) if tiles else linear(x, lp["wqkv"], lp["bqkv"], 3 * D, D)
            att = torch.empty(R, D, **f32)
            gated = cfg.gru_rel_pos and relb is not None
            lib.call("sed_attention_relpos", qkv.data_ptr(), relb.data_ptr() if relb is not None else None,
                     a.grep_linear.weight.data_ptr() if gated else None, a.grep_linear.bias.data_ptr() if gated else None,
                     lp["grep_a"].data_ptr() if gated else None, att.data_ptr(), B, T, H, D // H, st)
            o = linear(att, a.out_proj.weight, a.out_proj.bias, D, D)
            x = layernorm(o, x, alpha, lyr.self_attn_layer_norm, D, image=ximg2)
            h2 = None
            if ffn_tiles:
                linear_tiles(ximg2, lyr.fc1.weight, lyr.fc1.bias, Fd, D, act=1, out_image=himg)
                if LINEAR_TILES_KSPLIT and (Fd // 16) % 2 == 0:
                    # N = 768: 279 tiles on 256 CUs would be two rounds; two K halves = 558 items = three half-rounds, the partial sums
                    # added by the LayerNorm that follows
                    hh = linear_tiles(himg, lyr.fc2.weight, lyr.fc2.bias, D, Fd, split2=True)
                    h, h2 = hh[0], hh[1]
                else:
                    h = linear_tiles(himg, lyr.fc2.weight, lyr.fc2.bias, D, Fd)
            else:
                h = linear(x, lyr.fc1.weight, lyr.fc1.bias, Fd, D, act=1)
                h = linear(h, lyr.fc2.weight, lyr.fc2.bias, D, Fd)
            x = layernorm(h, x, alpha, lyr.final_layer_norm, D, image=ximg, x2=h2)     # (the last layer's image is written and not read)
            if taps is not None:
                taps["layer%d" % len([k for k in taps if k.startswith("layer")])] = x.view(B, T, D)
        return x.view(B, T, D), None


class BEATsModel(nn.Module):
    """BEATs.py:205-223: loads `{"cfg", "model"}` from `cfg_path`; forward(x (B, N) waveforms) -> global / frame embeddings."""

    def __init__(self, cfg_path=None, checkpoint=None):
        super().__init__()
        if checkpoint is None:
            checkpoint = torch.load(cfg_path, map_location="cpu", weights_only=False)
        cfg = BEATsConfig(checkpoint["cfg"])
        model = BEATs(cfg)
        model.load_state_dict(checkpoint["model"])
        self.model = model
        self.ckpt = checkpoint
        self.eval()

    def forward(self, x):
        features = self.model.extract_features(x)[0]
        return {"global": features.mean(dim=1).float(), "frame": features.transpose(1, 2).float()}
