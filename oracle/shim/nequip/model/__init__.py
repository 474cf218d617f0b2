"""ORACLE SHIM: `model_builder` -- handles seed / model_dtype / compile_mode at the outermost call."""
import functools

import torch

from ..utils.dtype import torch_default_dtype

_DEPTH = [0]


def model_builder(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if _DEPTH[0] > 0:
            return fn(*args, **kwargs)
        seed = kwargs.pop("seed", None)
        model_dtype = kwargs.pop("model_dtype", "float32")
        kwargs.pop("compile_mode", None)
        dtype = {"float32": torch.float32, "float64": torch.float64}[model_dtype]
        _DEPTH[0] += 1
        try:
            with torch_default_dtype(dtype):
                if seed is not None:
                    torch.manual_seed(seed)
                return fn(*args, **kwargs)
        finally:
            _DEPTH[0] -= 1

    return wrapper
