"""The mel kernel as a hipGraph node beside the BiGRU tails, replayed N times, every output compared with the solo launch
(tests/parity_cases.case_mel_in_graph_beside_tails).  python tools/mel_graph_race.py [library.so | -] [replays] [wave]
`wg`: the round-1..4 workgroup-per-frame kernel (mel_wave = 2) instead of the default wave-per-frame kernel.  Variant builds:
ONLY=sed_mel.hip python tools/build_variant.py w8 -DMEL_WAVES=8 -> tools/_libsed_w8.so.  (The multi-frame form that fails intermittently
is not in the library any more: tools/mel_repro/.)"""
import sys
sys.path.insert(0, ".")
from desed_task_amd import _lib
_lib.use_library(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None, is_emulator=False)
from tests import parity_cases as P
if "wg" in sys.argv[3:]:
    _lib.set_tuning("mel_wave", 2)
try:
    beside = [a for a in sys.argv[3:] if a in ("tails", "matmul", "rnn", "cnn", "none", "gemm", "gru", "storm")]
    n = P.case_mel_in_graph_beside_tails("cuda", replays=int(sys.argv[2]) if len(sys.argv) > 2 else 400, beside=beside[0] if beside else "tails")
    print(sys.argv[1:], "replays", n, "bad 0")
except AssertionError as e:
    print(sys.argv[1:], "FAILED", str(e)[:400])
