"""desed_task.utils: schedulers / scaler served by desed_task_amd.utils; the package's own exports
(`ManyHotEncoder`, `ExponentialWarmup`) are kept, the encoder coming from the reference when it is installed."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from .schedulers import ExponentialWarmup  # noqa: E402,F401

try:  # the label encoder is outside the hot path: the reference's own module, if present
    from .encoder import ManyHotEncoder  # noqa: F401
except ImportError:  # pragma: no cover
    pass
