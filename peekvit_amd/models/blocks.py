"""Building blocks with the reference's names and parameter layout (reference models/blocks.py).

`MLP` and `SelfAttention` own stock `nn.Linear` / `nn.MultiheadAttention` submodules purely as PARAMETER
CONTAINERS (state-dict keys `mlp.fc1.weight`, `self_attention.self_attention.in_proj_weight`, ... must match
the reference).  On the MI355X path their arithmetic is executed by
peekvit_amd.engine.block_forward (hand-written HIP kernels); the `forward` methods below are the stock-op
composite used for CPU tensors and while autograd records.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class SigmoidWithTemp(nn.Module):
    """sigmoid(x / temp + bias)  (reference models/blocks.py:62-69)."""

    def __init__(self, bias: float = 0, temp: float = 1.0):
        super().__init__()
        self.temp, self.bias = temp, bias

    def forward(self, x):
        return torch.sigmoid(x / self.temp + self.bias)


class GumbelSigmoid(nn.Module):
    """Straight-through Gumbel sigmoid gate (reference models/blocks.py:29-57).  RNG-driven research knob,
    outside the accelerated path: kept only so `gate_type='gumbel'` models construct and run on stock ops."""

    def __init__(self, hard: bool = True, temp: float = 1.0, bias: float = 0.0):
        super().__init__()
        self.hard, self.temp, self.bias = hard, temp, bias

    def forward(self, x):
        if not self.training:
            return torch.round(torch.sigmoid(x))
        noise = -torch.empty_like(x).exponential_().log()
        soft = torch.sigmoid((x + noise) / self.temp + self.bias)
        if not self.hard:
            return soft
        return torch.round(soft) - soft.detach() + soft


class MLP(nn.Module):
    """fc2(gelu_erf(fc1(x)))  (reference models/blocks.py:74-84)."""

    def __init__(self, hidden_dim, mlp_dim):
        super().__init__()
        self.fc1 = nn.Linear(hidden_dim, mlp_dim)
        self.fc2 = nn.Linear(mlp_dim, hidden_dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class SelfAttention(nn.Module):
    """Wrapper around nn.MultiheadAttention(batch_first=True) (reference models/blocks.py:88-95).
    The head-averaged attention map the reference computes with need_weights=True is discarded there;
    the composite below does not request it and the HIP kernel never forms it."""

    def __init__(self, input_dim, num_heads, dropout=0.0):
        super().__init__()
        self.self_attention = nn.MultiheadAttention(input_dim, num_heads, batch_first=True, dropout=dropout)

    def forward(self, x, attn_mask=None, key_padding_mask=None):
        out, _ = self.self_attention(x, x, x, need_weights=False, attn_mask=attn_mask,
                                     key_padding_mask=key_padding_mask)
        return out
