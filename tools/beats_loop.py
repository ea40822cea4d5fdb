"""The BEATs extractor in a loop (power / clock sampling: tools/power_probe.sh "python tools/beats_loop.py 400").  ZERO=1: all-zero weights."""
import os, sys, torch
sys.path.insert(0, ".")
from desed_task_amd.beats import BEATs, BEATsConfig
CFG = dict(input_patch_size=16, embed_dim=512, conv_bias=False, encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
           encoder_attention_heads=12, activation_fn="gelu", layer_norm_first=False, deep_norm=True, conv_pos=128, conv_pos_groups=16,
           relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True, dropout=0.0, attention_dropout=0.0,
           encoder_layerdrop=0.0)
torch.manual_seed(0)
model = BEATs(BEATsConfig(CFG)).cuda().eval()
audio = 0.1 * torch.randn(48, 160000, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
with torch.no_grad():
    for _ in range(3): model.extract_features(audio)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): model.extract_features(audio)
    e1.record(); torch.cuda.synchronize()
print("%d extractor passes, %.2f ms each" % (n, e0.elapsed_time(e1) / n))
