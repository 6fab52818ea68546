import torch
import torch.nn as nn
from itertools import repeat
def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else tuple(repeat(x, 2))
class DropPath(nn.Identity):
    def __init__(self, p=0.0):
        super().__init__()
def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)
