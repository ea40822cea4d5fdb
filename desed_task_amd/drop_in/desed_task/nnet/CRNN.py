"""`from desed_task.nnet.CRNN import CRNN` (train_sed.py:14) -> the HIP-kernel CRNN."""
from desed_task_amd.nnet.CRNN import CRNN  # noqa: F401
