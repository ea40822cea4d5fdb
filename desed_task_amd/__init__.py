"""desed_task_amd: MI355X-native (gfx950) mel + CRNN mean-teacher training path of DESED_task.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the
hot path runs in hand-written HIP kernels behind the C-ABI of include/sed_hip.h (libsed_hip.so).
"""
__version__ = "0.1.0"
