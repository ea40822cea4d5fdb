"""`from desed_task.utils.schedulers import ExponentialWarmup` (train_sed.py:16)."""
from desed_task_amd.utils.schedulers import BaseScheduler, ExponentialWarmup  # noqa: F401
