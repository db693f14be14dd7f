"""Residual-gated ViT behind the reference's interface (reference models/residualvit.py).

Each residual block scores every image token with a learned gate and multiplies the token (and the
block's LayerNorm / attention outputs) by the resulting soft mask; the sequence length never shrinks
(models/residualvit.py:197-260).  The MI355X path covers the configuration every shipped
residual*.yaml uses: eval mode, `gate_type='sigmoid'`, `skip='attention+mlp'`, `add_budget_token=
'learnable'`, `add_input=False` - one gate kernel produces the mask, the masked residual stream and the
row-scale vector that the LayerNorm and out-proj epilogues apply.  Everything else (gumbel gate, the
'attention' / 'mlp' skip modes, sampled budgets in training) runs on the stock-op composite.
"""
from __future__ import annotations

from abc import ABC
from typing import List, Literal, Optional, Union

import torch
from torch import nn
import torch.nn.functional as F

from .. import engine, ops, train_engine
from .blocks import MLP, GumbelSigmoid, SelfAttention, SigmoidWithTemp
from .vit import _ViTBase, _make_layers


class ResidualModule(ABC, nn.Module):
    """Marker base class: callers find gated blocks with isinstance(m, ResidualModule)
    (reference models/residualvit.py:17-18, utils/utils.py:111-133)."""


class ResidualGate(nn.Module):
    """Per-token scalar gate (reference models/residualvit.py:21-74)."""

    def __init__(self, hidden_dim, threshold: Union[float, str] = 0.5, temp=1.0, gate_type='gumbel',
                 sigmoid_bias: float = 10.0):
        super().__init__()
        self.projection = nn.Linear(hidden_dim, 1)
        self.temp, self.gate_type, self.sigmoid_bias = temp, gate_type, sigmoid_bias
        if gate_type == 'gumbel':
            self.gate = GumbelSigmoid(hard=True, temp=temp, bias=sigmoid_bias)
        elif gate_type == 'sigmoid':
            self.gate = SigmoidWithTemp(temp=temp, bias=sigmoid_bias)
        else:
            raise ValueError(f'Unknown gate type {gate_type}')
        if gate_type == 'gumbel' and threshold != 0.5:
            raise ValueError(f'Gumbel gate cannot have a threshold different from 0.5')
        if isinstance(threshold, float):
            self.threshold = threshold
        elif threshold == 'learnable':
            self.threshold = nn.Parameter(torch.tensor(0.5))

    def forward(self, x, budget: float = None, threshold: float = None):
        assert x.dim() == 3, f'Expected (batch_size, seq_length, hidden_dim) got {x.shape}'
        assert budget is None or threshold is None, 'Cannot specify both budget and threshold'
        mask = self.gate(self.projection(x))
        if self.gate_type != 'sigmoid':
            assert budget is None, 'Gumbel gate does not support budget'
            return mask
        if budget is not None:
            cut = 1 - budget
        elif threshold is not None:
            cut = threshold
            self.threshold = threshold
        else:
            cut = self.threshold
        return F.relu(mask - cut)


class ResidualViTBlock(ResidualModule):
    """reference models/residualvit.py:81-273."""

    LN_EPS = 1e-6

    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float, attention_dropout: float,
                 temp: float = 1.0, add_input: bool = False, num_class_tokens: int = 1, num_registers: int = 0,
                 skip: Literal['attention', 'mlp', 'attention+mlp', 'none'] = None,
                 gate_type: Literal['gumbel', 'sigmoid'] = 'gumbel', gate_bias: float = 10.0,
                 gate_threshold: float = 0.5, budget_token: Union[bool, List, Literal['learnable']] = False):
        super().__init__()
        self.num_heads, self.hidden_dim, self.mlp_dim = num_heads, hidden_dim, mlp_dim
        self.budget_token = budget_token
        self.num_special_tokens = num_class_tokens + num_registers
        self.gate_type, self.skip = gate_type, skip
        self._p_drop = max(float(dropout), float(attention_dropout))
        if skip in {'attention', 'mlp', 'attention+mlp'}:
            self.temp, self.add_input = temp, add_input
            self.residual_gate = ResidualGate(hidden_dim, threshold=gate_threshold, temp=temp, gate_type=gate_type,
                                              sigmoid_bias=gate_bias)
        self.ln_1 = nn.LayerNorm(hidden_dim, eps=self.LN_EPS)
        self.self_attention = SelfAttention(hidden_dim, num_heads, dropout=attention_dropout)
        self.dropout = nn.Dropout(dropout)
        self.ln_2 = nn.LayerNorm(hidden_dim, eps=self.LN_EPS)
        self.mlp = MLP(hidden_dim=hidden_dim, mlp_dim=mlp_dim)
        if self.budget_token == 'learnable':
            self.budget_token_gate = nn.Linear(hidden_dim, 1)

    # -- helpers ---------------------------------------------------------------------------------------------
    def _split(self, tokens):
        special, img = tokens[:, :self.num_special_tokens], tokens[:, self.num_special_tokens:]
        btok = None
        if self.budget_token:
            btok, img = img[:, -1:], img[:, :-1]
        return special, img, btok

    def _hip_gated(self, input: torch.Tensor) -> bool:
        return (self.skip == 'attention+mlp' and self.gate_type == 'sigmoid' and self.budget_token == 'learnable'
                and not self.add_input and self.num_special_tokens == 1
                and engine.backend_for(input, self, self._p_drop) == "hip")

    # -- skip modes (stock-op composites) ---------------------------------------------------------------------
    def forward_skip_attention(self, input: torch.Tensor):
        special, img, btok = self._split(input)
        self.mask = self.residual_gate(img, budget=btok.mean() if self.budget_token else None)
        gated = torch.cat([special, self.mask * img], dim=1)
        mid = self.dropout(self.self_attention(self.ln_1(gated))) + input
        return self.mlp(self.ln_2(mid))

    def forward_skip_mlp(self, input: torch.Tensor):
        mid = self.dropout(self.self_attention(self.ln_1(input))) + input
        special, img, btok = self._split(mid)
        self.mask = self.residual_gate(img, budget=btok.mean() if self.budget_token else None)
        pieces = [special, self.mask * img] + ([btok] if self.budget_token else [])
        y = self.mlp(self.ln_2(torch.cat(pieces, dim=1)))
        if self.add_input:
            y = y + torch.cat([torch.zeros_like(special), img * (1 - self.mask)], dim=1)
        return y

    def _hip_gated_train(self, input: torch.Tensor) -> bool:
        return (self.gate_type == 'sigmoid' and self.budget_token == 'learnable' and not self.add_input and self.num_special_tokens == 1
                and input.shape[1] >= 3 and train_engine.train_eligible(input, self, self._p_drop)
                and train_engine.supported(self.hidden_dim, self.num_heads, input.shape[1]))

    def forward_skip_attention_mlp(self, input: torch.Tensor):
        if self._hip_gated(input):
            return engine.run_guarded(self, input, lambda: self._hip_gated_block(input))
        if self._hip_gated_train(input):
            # training on the MI355X kernels: gate + masking (GateFn) and the masked block (MaskedBlockFn), both with hand-written backward;
            # block.mask is a view of the gate's output and stays differentiable for the auxiliary mask losses (utils/losses.py)
            masked, row_scale, thr, h1 = train_engine.gate_forward_train(self, input)
            self.mask = train_engine.enter(row_scale)[:, self.num_special_tokens:-1].unsqueeze(-1)     # (auxiliary mask losses: their gradient enters the scaled chain here)
            self.residual_gate.threshold = thr.view(-1, 1, 1)                 # what ResidualGate.forward leaves behind (residualvit.py:66)
            return train_engine.masked_block_forward_train(self, masked, row_scale, h1=h1)
        special, img, btok = self._split(input)
        budget, threshold = None, None
        if self.budget_token:
            budget = btok.mean()
        if self.budget_token == 'learnable':
            threshold, budget = torch.sigmoid(self.budget_token_gate(btok)), None
        self.mask = self.residual_gate(img, budget=budget, threshold=threshold)
        pieces = [special, self.mask * img] + ([btok] if self.budget_token else [])
        ones = torch.ones((self.mask.size(0), 1, self.mask.size(2)), device=self.mask.device)
        y = self.plain_forward(torch.cat(pieces, dim=1), mask=torch.cat([ones, self.mask, ones], dim=1))
        if self.add_input:
            y = y + torch.cat([torch.zeros_like(special), img * (1 - self.mask)], dim=1)
        return y

    def _hip_gated_block(self, input: torch.Tensor, rows: int = 0):
        x = input if input.is_contiguous() else input.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        gate, bgate = self.residual_gate.projection, self.budget_token_gate
        thr = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        # the gate kernel holds every row in registers: it also emits row_scale * LN1(masked row), the block's first step (residualvit.py:251)
        h1 = None
        if engine._PRECISION != "bf16x3" and engine._GATE_LN1 and not engine.layer_is_hybrid():     # (a hybrid layer normalises the masked tokens itself, in split precision)
            engine._check_ln_range(self.ln_1)
            h1 = engine.workspace.get("h", (x.shape[0] * x.shape[1], x.shape[2]), engine._lib.operand_dtype(), x.device)
        # with h1 from the gate and a tile GEMM for out-proj the masked tokens themselves are never needed: the out-proj epilogue computes
        # row_scale * (x + branch) from the unmasked rows (pv_gemm_args.res_scaled)
        no_masked = h1 is not None and engine._GATE_NO_MASKED and not engine._ln_fusable(x.shape[2], x.shape[2]) and not engine.layer_mlp_is_hybrid()      # (a split MLP half reads the masked tokens' x1)
        masked = None if no_masked else torch.empty_like(x)
        f32 = engine._f32
        self.mask, row_scale = ops.residual_gate(x, masked, f32(gate.weight), f32(gate.bias), f32(bgate.weight), f32(bgate.bias),
                                                 self.residual_gate.temp, self.residual_gate.sigmoid_bias, thr_out=thr,
                                                 ln=None if h1 is None else (f32(self.ln_1.weight), f32(self.ln_1.bias), self.ln_1.eps, h1))
        self.residual_gate.threshold = thr.view(-1, 1, 1)           # what ResidualGate.forward leaves behind (residualvit.py:66; utils.py:131)
        if rows:          # the rows read there (class tokens) carry scale 1: masked == unmasked
            return engine.block_forward_rows(self, x if no_masked else masked, self.ln_1.eps, rows, row_scale=row_scale, h1=h1)
        if no_masked:
            return engine.block_forward(self, x, self.ln_1.eps, row_scale=row_scale, h1=h1, res_scaled=True)
        return engine.block_forward(self, masked, self.ln_1.eps, row_scale=row_scale, h1=h1)

    def _pv_forward_rows(self, input: torch.Tensor, nq: int):
        """engine.run_layers, last block of a model forward (inference): the gate still sees every token and `self.mask` is the full mask,
        the block's output is computed for the class-token rows only."""
        if type(self) is not ResidualViTBlock or input.dim() != 3 or not engine.rows_only_ok(self) or nq > self.num_special_tokens:
            return None
        if self.skip == 'attention+mlp' and nq == 1 and self._hip_gated_train(input):
            # training: gate over every token (block.mask stays the full, differentiable mask), masked block on the class-token row
            masked, row_scale, thr, h1 = train_engine.gate_forward_train(self, input)
            self.mask = train_engine.enter(row_scale)[:, self.num_special_tokens:-1].unsqueeze(-1)     # (auxiliary mask losses: their gradient enters the scaled chain here)
            self.residual_gate.threshold = thr.view(-1, 1, 1)
            return train_engine.block_forward_rows_train(self, masked, mask=row_scale, h1=h1)
        if self.skip == 'attention+mlp' and self._hip_gated(input):
            return engine.run_guarded(self, input, lambda: self._hip_gated_block(input, rows=nq))
        if self.skip not in ('attention', 'mlp', 'attention+mlp') and engine.backend_for(input, self, self._p_drop) == "hip":
            return engine.run_guarded(self, input, lambda: engine.block_forward_rows(self, input, self.ln_1.eps, nq))
        return None

    def plain_forward(self, input: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """Masked pre-LN block: the mask multiplies LN1's output, the attention branch and LN2's output
        (reference models/residualvit.py:249-260)."""
        if mask is None:
            if train_engine.train_eligible(input, self, self._p_drop) and train_engine.supported(self.hidden_dim, self.num_heads, input.shape[1]):
                return train_engine.block_forward_train(self, input)
            if engine.backend_for(input, self, self._p_drop) == "hip":
                return engine.run_guarded(self, input, lambda: engine.block_forward(self, input, self.ln_1.eps))
            mask = torch.tensor(1.0, device=input.device)
        elif (mask.dim() == 3 and mask.shape[:2] == input.shape[:2] and train_engine.train_eligible(input, self, self._p_drop)
              and train_engine.supported(self.hidden_dim, self.num_heads, input.shape[1])):
            # training on the MI355X kernels: masked block forward + backward incl. the gradient of the mask (train_engine.MaskedBlockFn)
            return train_engine.masked_block_forward_train(self, input, mask.to(input.device))
        mask = mask.to(input.device)
        mid = self.dropout(mask * self.self_attention(mask * self.ln_1(input))) + input
        return mid + self.mlp(mask * self.ln_2(mid))

    def forward(self, input: torch.Tensor):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        tp = train_engine.current_pass()
        if tp is not None and tp.scaled and not self._trains_on_hip(input):
            # a stock-op block in the middle of a loss-scaled fp16 training pass (skip modes 'attention' / 'mlp', gumbel gates, ...): its
            # parameters must see TRUE gradients - the gradient leaves the scaled chain behind the block and re-enters it in front
            with train_engine.stock_region():
                return train_engine.leave(self._dispatch(train_engine.enter(input, tp=tp)), tp=tp)
        return self._dispatch(input)

    def _trains_on_hip(self, input: torch.Tensor) -> bool:
        if self.skip == 'attention+mlp':
            return self._hip_gated_train(input)
        if self.skip in ('attention', 'mlp'):
            return False
        return train_engine.train_eligible(input, self, self._p_drop) and train_engine.supported(self.hidden_dim, self.num_heads, input.shape[1])

    def _dispatch(self, input: torch.Tensor):
        if self.skip == 'attention':
            return self.forward_skip_attention(input)
        if self.skip == 'mlp':
            return self.forward_skip_mlp(input)
        if self.skip == 'attention+mlp':
            return self.forward_skip_attention_mlp(input)
        return self.plain_forward(input)


class ResidualViTEncoder(nn.Module):
    """reference models/residualvit.py:278-348: pos-embedding covers every token but the trailing budget token."""

    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float,
                 attention_dropout: float, residual_layers: Optional[List] = None, add_input: bool = False,
                 num_class_tokens: int = 1, num_registers: int = 0, gate_type: Literal['gumbel', 'sigmoid'] = 'gumbel',
                 gate_temp: float = 1.0, gate_bias: float = 10.0, gate_threshold: float = 0.5,
                 budget_token: Union[bool, List, Literal['learnable']] = False):
        super().__init__()
        self.num_layers = num_layers
        self.num_class_tokens, self.num_registers = num_class_tokens, num_registers
        self.num_special_tokens = num_class_tokens + num_registers
        self.budget_token = budget_token
        self.num_budget_tokens = 0 if not budget_token else 1
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = nn.Dropout(dropout)
        self.layers = _make_layers(
            lambda i: ResidualViTBlock(num_heads, hidden_dim, mlp_dim, dropout, attention_dropout,
                                       skip=residual_layers[i], add_input=add_input,
                                       num_class_tokens=num_class_tokens, num_registers=num_registers,
                                       gate_type=gate_type, temp=gate_temp, gate_bias=gate_bias,
                                       gate_threshold=gate_threshold, budget_token=budget_token), num_layers)
        self.ln = nn.LayerNorm(hidden_dim)

    def forward(self, input: torch.Tensor, _pos_added: bool = False, _rows: int = 0):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        if _pos_added:
            return engine.run_layers(self.layers, input, last_rows=_rows)      # gated blocks publish no _pv_plain_ln1: they normalise themselves
        if self.budget_token:
            body, btok = input[:, :-self.num_budget_tokens], input[:, -self.num_budget_tokens:]
            input = torch.cat([body + self.pos_embedding, btok], dim=1)
        else:
            input = input + self.pos_embedding
        return self.ln(self.layers(self.dropout(input)))


class ResidualVisionTransformer(_ViTBase):
    """reference models/residualvit.py:352-694."""

    def __init__(self, image_size: int, patch_size: int, num_layers: int, num_heads: int, hidden_dim: int,
                 mlp_dim: int, dropout: float = 0.0, attention_dropout: float = 0.0, num_classes: int = 1000,
                 representation_size: Optional[int] = None, num_registers: int = 0,
                 residual_layers: Optional[List] = None, add_input: bool = False, num_class_tokens: int = 1,
                 gate_type: Literal['gumbel', 'sigmoid'] = 'gumbel', gate_temp: float = 1.0, gate_bias: float = 10.0,
                 gate_threshold: float = 0.5,
                 add_budget_token: Union[bool, List, Literal['learnable', 'learnable_interpolate']] = False,
                 budget_interval: Optional[List] = (0, 1), torch_pretrained_weights: Optional[str] = None,
                 timm_pretrained_weights: Optional[List] = None, remove_layers: List[int] = []):
        super().__init__()
        seq_length = self._init_stem(image_size, patch_size, hidden_dim, mlp_dim, dropout, attention_dropout,
                                     num_classes, representation_size, num_heads, num_registers, num_class_tokens)
        self.add_budget_token = add_budget_token
        self.current_budget = None
        self.gate_temp, self.gate_bias = gate_temp, gate_bias
        self.budget_interval = budget_interval
        self.residual_layers = residual_layers or ['attention+mlp'] * num_layers
        if num_registers > 0:
            self.register_tokens = nn.Parameter(torch.zeros(1, num_registers, hidden_dim))
            seq_length += num_registers
        self.num_special_tokens = num_class_tokens + num_registers
        self.encoder = ResidualViTEncoder(seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout,
                                          attention_dropout, residual_layers=self.residual_layers,
                                          add_input=add_input, gate_type=gate_type, gate_temp=gate_temp,
                                          gate_bias=gate_bias, gate_threshold=gate_threshold,
                                          budget_token=add_budget_token)
        self.seq_length = seq_length
        if self.add_budget_token:
            self.num_budget_tokens = 1
            if self.add_budget_token in ('learnable', 'learnable_interpolate'):
                self.learnable_budget_token_1 = nn.Parameter(torch.randn(1, 1, hidden_dim))
            if self.add_budget_token == 'learnable_interpolate':
                self.learnable_budget_token_2 = nn.Parameter(torch.randn(1, 1, hidden_dim))
                self.num_budget_tokens = 2
        self._init_head()
        self.load_weights(torch_pretrained_weights, timm_pretrained_weights)
        if remove_layers:
            self.remove_layers(remove_layers)

    # -- budget token ----------------------------------------------------------------------------------------
    def _sample_budget(self, n):
        """One budget per image, uniform in budget_interval (reference models/residualvit.py:541-550; its
        list branch calls `random.choice` on a function and cannot run - SURVEY appendix A.3 - so list-valued
        add_budget_token is rejected here instead of failing with AttributeError)."""
        if isinstance(self.add_budget_token, (list, tuple)):
            raise NotImplementedError("list-valued add_budget_token is broken in the reference and unsupported")
        if isinstance(self.add_budget_token, float):
            return torch.tensor(self.add_budget_token)
        lo, hi = self.budget_interval
        return torch.rand(n) * (hi - lo) + lo

    def _add_budget_token(self, x, wrap=None):
        """Append the budget token row(s) (reference models/residualvit.py:552-585)."""
        n = x.shape[0]
        if self.training:
            self.current_budget = self._sample_budget(n).to(x.device)
        else:
            assert self.current_budget is not None, 'Budget token not set. Call set_budget() before forward() to evaluate the model on a chosen budget.'
        if self.add_budget_token == 'learnable':
            scale = self.current_budget.unsqueeze(-1).unsqueeze(-1)
            extra = self.learnable_budget_token_1.expand(n, -1, -1) * scale
        elif self.add_budget_token == 'learnable_interpolate':
            extra = (self.learnable_budget_token_1.expand(n, -1, -1) * self.current_budget
                     + self.learnable_budget_token_2.expand(n, -1, -1) * (1 - self.current_budget))
        else:
            extra = torch.empty((n, 1, self.hidden_dim), device=x.device).fill_(self.current_budget)
        return torch.cat([x, wrap(extra) if wrap is not None else extra], dim=1)

    def forward(self, x: torch.Tensor):
        self._check_image(x)
        if x.shape[0] == 0:                    # a batch of zero images (the reference's nn.MultiheadAttention raises on it): empty logits
            return x.new_zeros((0, self.num_classes), dtype=torch.float32)
        hip = engine.backend_for(x, self, max(self.dropout, self.attention_dropout)) == "hip"
        if hip and self.add_budget_token in (False, None, 'learnable') and not self.training:
            btok, budget = None, 0.0
            if self.add_budget_token == 'learnable':
                assert self.current_budget is not None, 'Budget token not set. Call set_budget() before forward() to evaluate the model on a chosen budget.'
                cached = getattr(self, "_pv_budget", None)      # set_budget's value as a host float: no device read (a sync; illegal under graph capture)
                budget = cached[1] if cached is not None and cached[0] is self.current_budget else float(self.current_budget)
                btok = self.learnable_budget_token_1.detach().view(-1)
            body = lambda xs: engine.pool_and_head(self, engine.call_module(self.encoder, engine.embed_tokens(self, xs, btok, budget), _pos_added=True,
                                                                            _rows=self.num_class_tokens))
            return engine.run_guarded(self, x, lambda: body(x), probe=body, probe_key=budget)
        if (self.training and train_engine.train_eligible(x, self, max(self.dropout, self.attention_dropout))
                and train_engine.supported(self.hidden_dim, self.num_heads, self.seq_length + (1 if self.add_budget_token else 0))):
            # training on the MI355X kernels end to end: patch embedding (+ class tokens, + pos_embedding) and its backward are the
            # ViT's EmbedFn; the budget token row is appended behind it (it carries no positional embedding, residualvit.py:338-345);
            # the gated blocks dispatch themselves (MaskedBlockFn); final LayerNorm + head on the class rows only
            def train_body():
                tokens = train_engine.embed_tokens_train(self, x)
                if self.add_budget_token:
                    # the budget-token row is built by stock ops from a parameter: its gradient LEAVES the scaled chain there
                    tokens = self._add_budget_token(tokens, wrap=train_engine.leave)
                return train_engine.pool_and_head_train(self, self.encoder(tokens, _pos_added=True, _rows=self.num_class_tokens))
            with engine.on_device(x):
                return train_engine.model_forward_train(self, x, train_body)
        tokens = self._composite_tokens(x)
        if self.add_budget_token:
            tokens = self._add_budget_token(tokens)
        return self._composite_head(self.encoder(tokens))

    def set_budget(self, budget: float):
        if self.training:
            raise ValueError('You cannot set the budget during training in this model. This model has a learnable budget so you have to set it at the beginning of the training and then sample it during training. Use the add_budget_token parameter to specify the budget sampling strategy.')
        self.current_budget = torch.tensor(budget, device=self.class_tokens.device)
        object.__setattr__(self, "_pv_budget", (self.current_budget, float(budget)))
