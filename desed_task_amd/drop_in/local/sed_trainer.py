"""`from local.sed_trainer import SEDTask4` (train_sed.py:19) -> the HIP-kernel mean-teacher trainer."""
from desed_task_amd.sed_trainer import SEDTask4  # noqa: F401
