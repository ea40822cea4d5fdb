"""desed_task.nnet with CRNN / CNN / RNN served by desed_task_amd.nnet; other modules come from the reference."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
