"""desed_task.utils.scaler -> TorchScaler on the HIP min/max kernels."""
from desed_task_amd.utils.scaler import TorchScaler  # noqa: F401
