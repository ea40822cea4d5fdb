import time, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from tests import parity_cases as P
from oracle import sed_oracle as O
mel = P.make_mel()
audio = (0.1 * torch.randn(48, 160000)).cuda()
for _ in range(3): out = mel(audio)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = mel(audio)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("mel 48 clips: %.3f ms -> %.1f GB/s algorithmic" % (ms, 48 * 960512 / ms / 1e6))
