"""`from local.sed_trainer_pretrained import SEDTask4` (train_pretrained.py) -> the trainer with embeddings in the batch."""
from desed_task_amd.sed_trainer_pretrained import SEDTask4  # noqa: F401
