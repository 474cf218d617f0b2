import torch

_GLOBAL_DTYPE = torch.float64
