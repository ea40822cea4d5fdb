"""desed_task.nnet.CNN -> parameter containers of the HIP CNN encoder."""
from desed_task_amd.nnet.CNN import CNN, GLU  # noqa: F401
