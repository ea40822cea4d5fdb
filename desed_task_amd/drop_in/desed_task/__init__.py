"""Alias package: `desed_task` whose hot-path modules are the MI355X implementations (see ../README.md).  Everything not
aliased here is looked up in the reference's own `desed_task` package further down sys.path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
