import logging

from .dtype import torch_default_dtype, dtype_from_name  # noqa: F401


class RankedLogger(logging.LoggerAdapter):
    def __init__(self, name=__name__, rank_zero_only=False, extra=None):
        super().__init__(logging.getLogger(name), extra or {})
