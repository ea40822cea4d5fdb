"""Alias package for the 2024 recipe's `local/` directory: put this directory in FRONT of desed_task_amd/drop_in on PYTHONPATH
when running recipes/dcase2024_task4_baseline -- its `local.sed_trainer_pretrained` is the five-data-set trainer."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
