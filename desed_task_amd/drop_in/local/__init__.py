"""Alias package for the recipes' `local/` helper directory (a namespace directory in the reference): `local.sed_trainer` and
`local.sed_trainer_pretrained` are the MI355X trainers; every other `local.*` module (classes_dict, resample_folder, utils, ...)
is looked up in the recipe's own `local/` directory."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
