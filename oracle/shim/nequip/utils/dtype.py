import contextlib

import torch


@contextlib.contextmanager
def torch_default_dtype(dtype):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def dtype_from_name(name):
    return {"float32": torch.float32, "float64": torch.float64}[name] if isinstance(name, str) else name
