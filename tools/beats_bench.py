"""Throughput of the frozen BEATs extractor at the recipe's size: 48 clips of 10 s -> (48, 768, 496) embeddings,
iter3 configuration (12 layers), random weights.  Secondary workload: not the headline metric of bench.py."""
import json, sys, time
sys.path.insert(0, ".")
import torch
from desed_task_amd import _lib
from desed_task_amd import beats as _beats
from desed_task_amd.beats import BEATs, BEATsConfig
import os
if os.environ.get("LINEAR_TILES") == "0":       # A/B: the round-5 path of the q / k / v projection
    _beats.LINEAR_TILES = False
if os.environ.get("LINEAR_FORM"):                 # A/B: 5 = the loader-wave form of the tile Linear (sed_set_tuning linear_tiles)
    _lib.set_tuning("linear_tiles", int(os.environ["LINEAR_FORM"]))
if os.environ.get("LINEAR_TILES_KSPLIT") == "0":   # A/B: fc2 as one product per tile (two rounds of tiles)
    _beats.LINEAR_TILES_KSPLIT = False
if os.environ.get("LINEAR_TILES_FFN") == "0":   # A/B: only the q / k / v projection on the tile path
    _beats.LINEAR_TILES_FFN = False
CFG = dict(input_patch_size=16, embed_dim=512, conv_bias=False, encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
           encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False, deep_norm=True, conv_pos=128, conv_pos_groups=16,
           relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True, dropout=0.0, attention_dropout=0.0,
           encoder_layerdrop=0.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
if len(sys.argv) > 2:                   # A/B: another build of the C-ABI library (tools/_libsed_*.so)
    _lib.use_library(sys.argv[2], is_emulator=False)
torch.manual_seed(0)
model = BEATs(BEATsConfig(CFG)).cuda().eval()
audio = 0.1 * torch.randn(B, 160000, device="cuda")
for _ in range(2):
    feats, _ = model.extract_features(audio)
torch.cuda.synchronize()
lib = _lib.get(); orig = lib.call; rec = {}; work = {}
PEAK_BF16X3, PEAK_HBM = 2500.0 / 3.0, 8000.0        # TFLOP/s of algorithmic products at three bf16 MFMAs each; GB/s


def entry_work(name, a):
    """(bound, algorithmic FLOPs or bytes) of one launch from its C-ABI arguments (include/sed_hip.h)."""
    if name.startswith("sed_linear"):
        M, N, K = a[4:7]
        return "mfma", 2.0 * M * N * K
    if name == "sed_attention_relpos":
        B_, T, H, Dh = a[6:10]
        return "mfma", 4.0 * B_ * H * T * T * Dh
    if name.startswith("sed_posconv"):
        B_, T, D, K = a[4:8]
        return "mfma", 2.0 * B_ * T * D * K * (D // a[8])
    if name == "sed_layernorm":
        M, D = a[6:8]
        return "hbm", 4.0 * M * D * (3 if a[1] else 2)
    if name == "sed_layernorm_tiles":
        M, D = a[8:10]
        return "hbm", 4.0 * M * D * (3 + (1 if a[1] else 0) + (1 if a[2] else 0))
    if name == "sed_kaldi_fbank":
        B_, N = a[2:4]
        return "hbm", 4.0 * B_ * (N + (1 + (N - 400) // 160) * a[4])
    if name == "sed_patchify":
        B_, M, F = a[2:5]
        return "hbm", 8.0 * B_ * M * F
    return None, 0.0


def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record()
    rec.setdefault(name, []).append((e0, e1))
    b, w = entry_work(name, a)
    if b:
        work.setdefault(name, [b, 0.0])[1] += w
t0 = time.perf_counter()
n = 3
for _ in range(n):
    feats, _ = model.extract_features(audio)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
lib.call = timed
model.extract_features(audio); torch.cuda.synchronize()
lib.call = orig
per = {k: round(sum(a.elapsed_time(b) for a, b in v), 3) for k, v in rec.items()}
roof = {}
for k, (bound, w) in work.items():
    ach = w / (per[k] * 1e-3) / (1e12 if bound == "mfma" else 1e9)
    peak = PEAK_BF16X3 if bound == "mfma" else PEAK_HBM
    roof[k] = {"bound": bound, "launches": len(rec[k]), "ms": per[k], "achieved": round(ach, 1), "peak": round(peak, 1),
               "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(ach / peak, 4)}
flops = B * (496 * (2 * 256 * 512 + 2 * 512 * 768) + 12 * (496 * 2 * 768 * (3 * 768 + 768 + 2 * 3072) + 2 * 2 * 12 * 496 * 496 * 64) + 496 * 768 * 2 * 48 * 128)
print(json.dumps({"workload": "BEATs iter3 extractor, %d clips of 10 s -> (%d, 768, 496)" % (B, B), "ms_per_batch": round(dt * 1e3, 2),
                  "clips_per_s": round(B / dt, 1), "tflops_algorithmic": round(flops / dt / 1e12, 1), "ms_by_entry": per,
                  "roofline_by_entry": roof,
                  "roofline_note": "HIP events around every launch of one extractor pass; mfma entries: algorithmic FLOPs against 2500/3 = 833 "
                                   "TFLOP/s (split-bf16: three bf16 MFMAs per fp32-accurate product); hbm entries: algorithmic bytes against 8 TB/s",
                  "finite": bool(torch.isfinite(feats).all())}))
