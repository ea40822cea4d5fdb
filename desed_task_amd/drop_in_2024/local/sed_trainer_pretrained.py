"""`from local.sed_trainer_pretrained import SEDTask4` (dcase2024 train_pretrained.py) -> the five-data-set HIP trainer."""
from desed_task_amd.sed_trainer_pretrained_2024 import SEDTask4  # noqa: F401
