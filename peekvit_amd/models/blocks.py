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


class NoiseBlock(nn.Module):
    """Channel-noise module the evaluation harness splices between two encoder blocks (reference models/blocks.py:100-186;
    utils/utils.py:162-191 inserts it, validate/test.py:72-111 sweeps its value).  Stock PyTorch and RNG-driven, deliberately: it is a
    measurement knob, not part of the accelerated path - the blocks on either side keep running on the HIP kernels, and since this
    module exposes no `_pv_plain_ln1` the block in front of it fuses nothing into its epilogue (engine.run_layers).

    noise_type 'gaussian':   x + randn_like(x) * sqrt(mean(x^2, -1) / 10^(snr_db / 10)); snr_db == 0 adds nothing (the reference's
                             convention, not 0 dB); `std` is refused exactly as the reference refuses it.
    noise_type 'token_drop': int(prob * S) token positions, drawn once per call with torch.randperm (the host generator) and shared by
                             the whole batch, are zeroed.
    Lazily configured: snr / prob may be None until set_snr / set_prob / set_value is called (the sweep does that)."""

    def __init__(self, noise_type: str = "gaussian", snr=None, std=None, prob=None):
        super().__init__()
        self.noise_type, self.snr_db, self.std, self.prob = noise_type, snr, std, prob
        if not any([snr, std, prob]):
            print("Lazy initialization of noise block. Please set the noise parameters using set_snr, set_std or set_prob before using the block.")
        if std:
            raise ValueError("std is not supported anymore. Please use snr instead.")

    def forward_snr(self, x):
        # snr 0 means "no noise" in the reference, and it still DRAWS the noise and multiplies it by 0 (models/blocks.py:124-129): the
        # generator advances exactly as there, so a seeded noise sweep stays sample-for-sample comparable after a 0.0 entry
        std = (x.pow(2).mean(dim=-1, keepdim=True) / (10 ** (self.snr_db / 10))).sqrt() if self.snr_db != 0 else 0
        return x + torch.randn_like(x) * std

    def forward_std(self, x):
        return x + torch.randn_like(x) * self.std

    def forward_token_drop(self, x):
        if self.prob == 0:
            return x
        keep = torch.ones_like(x)
        keep[:, torch.randperm(x.shape[1])[: int(self.prob * x.shape[1])], :] = 0
        return x * keep

    @torch.no_grad()
    def forward(self, x):
        if self.snr_db is not None:
            return self.forward_snr(x)
        if self.std is not None:
            return self.forward_std(x)
        return self.forward_token_drop(x)

    def set_snr(self, snr: float):
        assert self.noise_type == "gaussian"
        self.snr_db, self.std, self.prob = snr, None, None

    def set_prob(self, prob: float):
        assert self.noise_type == "token_drop"
        self.snr_db, self.std, self.prob = None, None, prob

    def set_value(self, value: float):
        if self.noise_type == "gaussian":
            self.set_snr(value)
        else:
            self.set_prob(value)
