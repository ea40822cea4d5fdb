"""desed_task.nnet.RNN -> parameter container of the HIP BiGRU."""
from desed_task_amd.nnet.RNN import BidirectionalGRU  # noqa: F401
