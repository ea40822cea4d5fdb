from .CRNN import CRNN  # noqa: F401
