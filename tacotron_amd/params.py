"""Flat fp32 parameter buffer in the reference's TF variable order (layout owned by libtaco_hip.so)."""
from __future__ import annotations

import math

import torch

from . import lib


def glorot_limit(dims):
    if len(dims) == 1:
        fan_in = fan_out = dims[0]
    elif len(dims) == 2:
        fan_in, fan_out = dims
    else:
        rf = 1
        for d in dims[:-2]:
            rf *= d
        fan_in, fan_out = dims[-2] * rf, dims[-1] * rf
    return math.sqrt(6.0 / (fan_in + fan_out))


def init_kind(name):
    """TF-r1.2 default initialisers (SURVEY §8a footer): glorot-uniform kernels / embedding / attention_v,
    zero biases except GRU gate biases (1.0), BN gamma 1 / beta 0."""
    if name.endswith('/gates/bias') or name.endswith('/gamma'):
        return 'ones'
    if name.endswith('/bias') or name.endswith('/beta'):
        return 'zeros'
    return 'glorot'


class ParamBuffer:
    """One contiguous tensor + named views.  `flat` is what Adam / all-reduce / the C ABI see."""

    def __init__(self, shape: lib.TacoShape, device='cpu'):
        self.shape = shape
        self.table = lib.param_table(shape)
        self.numel = lib.param_count(shape)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.index = {name: (off, size, dims) for name, off, size, dims in self.table}

    def view(self, name, flat=None):
        off, size, dims = self.index[name]
        return (self.flat if flat is None else flat)[off:off + size].view(*dims)

    def names(self):
        return [t[0] for t in self.table]

    def init_(self, seed=0):
        g = torch.Generator(device='cpu').manual_seed(seed)
        host = torch.zeros(self.numel, dtype=torch.float32)
        for name, off, size, dims in self.table:
            kind = init_kind(name)
            if kind == 'glorot':
                lim = glorot_limit(dims)
                host[off:off + size] = (torch.rand(size, generator=g, dtype=torch.float64) * 2 - 1).mul_(lim).float()
            elif kind == 'ones':
                host[off:off + size] = 1.0
        self.flat.copy_(host)
        return self

    def load_dict_(self, d):
        """d: name -> array-like with the table's dims (e.g. an oracle parameter dict)."""
        host = torch.zeros(self.numel, dtype=torch.float32)
        for name, off, size, dims in self.table:
            host[off:off + size] = torch.as_tensor(d[name], dtype=torch.float32).reshape(-1)
        self.flat.copy_(host)
        return self

    def to_dict(self, flat=None):
        src = (self.flat if flat is None else flat).detach().cpu()
        return {name: src[off:off + size].view(*dims).numpy().copy() for name, off, size, dims in self.table}
