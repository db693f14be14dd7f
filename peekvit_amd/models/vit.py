"""Plain Vision Transformer behind the reference's interface (reference models/vit.py).

Same constructor kwargs (= Hydra YAML keys of configs/model/vit_*.yaml), same state-dict keys, same
`forward(x[B,3,R,R]) -> logits[B,num_classes]`, same assertion messages.  GPU tensors under
torch.no_grad() run on hand-written gfx950 kernels via peekvit_amd.engine; `encoder.layers` stays an
indexable / deletable nn.Sequential of self-contained blocks so the reference's model surgery
(utils/utils.py:177-189 add_noise, models/vit.py:302-315 remove_layers) keeps working.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
from torch import nn

from .. import engine, train_engine
from .blocks import MLP, SelfAttention


class ViTBlock(nn.Module):
    """Pre-LN encoder block: x = in + MHA(LN1(in)); out = x + MLP(LN2(x))  (reference models/vit.py:19-55)."""

    LN_EPS = 1e-5

    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float, attention_dropout: float):
        super().__init__()
        self.num_heads, self.hidden_dim, self.mlp_dim = num_heads, hidden_dim, mlp_dim
        self._p_drop = max(float(dropout), float(attention_dropout))
        self.ln_1 = nn.LayerNorm(hidden_dim, eps=self.LN_EPS)
        self.self_attention = SelfAttention(hidden_dim, num_heads, attention_dropout)
        self.dropout = nn.Dropout(dropout)
        self.ln_2 = nn.LayerNorm(hidden_dim, eps=self.LN_EPS)
        self.mlp = MLP(hidden_dim=hidden_dim, mlp_dim=mlp_dim)
        # engine hint: the LayerNorm the next block applies first (fused into fc2's epilogue); always a plain attribute, never a
        # registered submodule (engine.run_layers)
        object.__setattr__(self, "_pv_next_ln", None)
        object.__setattr__(self, "_pv_next_ranks", False)

    def _pv_plain_ln1(self) -> bool:
        """True when this block applies ln_1 directly to its input (so a producer may pre-compute it)."""
        return True

    def _pv_forward_rows(self, input: torch.Tensor, nq: int):
        """engine.run_layers, last block of a model forward: the output rows [0, nq) of every image only, or None (= run forward())."""
        if type(self) is not ViTBlock:
            return None
        return self._pv_rows(input, nq)

    def _pv_rows(self, input: torch.Tensor, nq: int):
        if input.dim() != 3 or not engine.rows_only_ok(self):
            return None
        if train_engine.train_eligible(input, self, self._p_drop):
            if nq == 1 and train_engine.supported(self.hidden_dim, self.num_heads, input.shape[1]):
                return train_engine.block_forward_rows_train(self, input)
            return None
        if engine.backend_for(input, self, self._p_drop) != "hip":
            return None
        return engine.run_guarded(self, input, lambda: engine.block_forward_rows(self, input, self.ln_1.eps, nq))

    def _composite(self, tokens: torch.Tensor) -> torch.Tensor:
        attn = self.dropout(self.self_attention(self.ln_1(tokens)))
        mid = attn + tokens
        return mid + self.mlp(self.ln_2(mid))

    def forward(self, input: torch.Tensor):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        if train_engine.train_eligible(input, self, self._p_drop) and \
                train_engine.supported(self.hidden_dim, self.num_heads, input.shape[1]):
            return train_engine.block_forward_train(self, input)          # autograd records: HIP forward + HIP backward
        if engine.backend_for(input, self, self._p_drop) == "hip":
            return engine.run_guarded(self, input, lambda: engine.block_forward(self, input, self.ln_1.eps, next_ln=self._pv_next_ln,
                                                                              next_ranks=self._pv_next_ranks))
        return self._composite(input)


def _make_layers(block_factory, num_layers: int) -> nn.Sequential:
    return nn.Sequential(*[block_factory(i) for i in range(num_layers)])


class ViTEncoder(nn.Module):
    """pos-embedding add, L blocks, final LayerNorm (reference models/vit.py:59-95)."""

    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int,
                 dropout: float, attention_dropout: float):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = nn.Dropout(dropout)
        self.layers = _make_layers(
            lambda i: ViTBlock(num_heads, hidden_dim, mlp_dim, dropout, attention_dropout), num_layers)
        self.ln = nn.LayerNorm(hidden_dim)

    def forward(self, input: torch.Tensor, _pos_added: bool = False, _rows: int = 0):
        """`_pos_added` is private to this package: the fused patch-embedding epilogue has already added
        pos_embedding, so the add (and the inactive dropout) is skipped.  `_rows` (private too): the caller reads rows [0, _rows) of
        every image only, the last block may return just those."""
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        if _pos_added:
            # MI355X path: the final LayerNorm is applied to the class-token rows only, by pool_and_head
            return engine.run_layers(self.layers, input, last_rows=_rows)
        return self.ln(self.layers(self.dropout(input + self.pos_embedding)))


class _ViTBase(nn.Module):
    """Shared stem/head plumbing of VisionTransformer, RankVisionTransformer, ResidualVisionTransformer
    (the reference repeats it per class: models/vit.py:122-199, models/rankvit.py:176-232,
    models/residualvit.py:416-500)."""

    def _init_stem(self, image_size, patch_size, hidden_dim, mlp_dim, dropout, attention_dropout, num_classes,
                   representation_size, num_heads, num_registers, num_class_tokens) -> int:
        torch._assert(image_size % patch_size == 0, "Input shape indivisible by patch size!")
        self.image_size, self.patch_size = image_size, patch_size
        self.hidden_dim, self.mlp_dim = hidden_dim, mlp_dim
        self.attention_dropout, self.dropout = attention_dropout, dropout
        self.num_classes, self.representation_size = num_classes, representation_size
        self.num_heads = num_heads
        self.num_registers, self.num_class_tokens = num_registers, num_class_tokens
        self.conv_proj = nn.Conv2d(in_channels=3, out_channels=hidden_dim, kernel_size=patch_size, stride=patch_size)
        self.class_tokens = nn.Parameter(torch.zeros(1, num_class_tokens, hidden_dim))
        return (image_size // patch_size) ** 2 + num_class_tokens

    def _init_head(self):
        # reference models/vit.py:186-194: zero head, truncated-normal stem with std sqrt(1/fan_in), zero stem bias
        self.head = nn.Linear(self.hidden_dim, self.num_classes)
        nn.init.zeros_(self.head.weight)
        nn.init.zeros_(self.head.bias)
        fan_in = 3 * self.patch_size * self.patch_size
        nn.init.trunc_normal_(self.conv_proj.weight, std=math.sqrt(1 / fan_in))
        nn.init.zeros_(self.conv_proj.bias)
        # what precision mode "auto" has concluded about this model's parameters (engine.GuardState) does not outlive them
        self.register_load_state_dict_post_hook(engine.reset_guard)

    def _check_image(self, x: torch.Tensor):
        # fp32 NCHW (the reference's contract) or, MI355X path only, the raw uint8 NHWC image (normalisation fused in-kernel)
        h, w = (x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[2], x.shape[3])
        torch._assert(h == self.image_size, f"Wrong image height! Expected {self.image_size} but got {h}!")
        torch._assert(w == self.image_size, f"Wrong image width! Expected {self.image_size} but got {w}!")

    def _process_input(self, x: torch.Tensor) -> torch.Tensor:
        """Stock-op patch embedding: [B,3,R,R] -> [B,Np,D] (reference models/vit.py:203-222)."""
        self._check_image(x)
        t = self.conv_proj(x)
        return t.flatten(2).transpose(1, 2)

    def _composite_tokens(self, x: torch.Tensor) -> torch.Tensor:
        t = self._process_input(x)
        n = t.shape[0]
        if self.num_registers > 0:
            t = torch.cat([self.register_tokens.expand(n, -1, -1), t], dim=1)
        return torch.cat([self.class_tokens.expand(n, -1, -1), t], dim=1)

    def _composite_head(self, enc_out: torch.Tensor) -> torch.Tensor:
        # class tokens are SUMMED (reference models/vit.py:242-243)
        return self.head(enc_out[:, 0:self.num_class_tokens].sum(dim=1))

    def load_weights(self, torch_pretrained_weights: Optional[str] = None, timm_pretrained_weights: Optional[List] = None):
        """Local torchvision / timm checkpoints (reference models/vit.py:251-299).  The download branches of
        the reference need network access and are not available; a name that is not a local file raises."""
        assert not (torch_pretrained_weights and timm_pretrained_weights), \
            "You cannot load weights from both torch and timm at the same time."
        src = torch_pretrained_weights if torch_pretrained_weights is not None else timm_pretrained_weights
        if src is None:
            return
        if not os.path.exists(str(src)):
            raise FileNotFoundError(f"pretrained weights {src!r}: only local checkpoint files are supported "
                                    "(no network); see peekvit_amd.models.adapters")
        from .adapters import adapt_timm_state_dict, adapt_torch_state_dict
        ckpt = torch.load(src, map_location="cpu")
        sd = ckpt["model"] if "model" in ckpt else ckpt.get("state_dict", ckpt)
        adapt = adapt_torch_state_dict if torch_pretrained_weights is not None else adapt_timm_state_dict
        self.load_state_dict(adapt(sd, num_classes=self.num_classes), strict=False)

    def remove_layers(self, remove_layers: List[int]):
        """Delete encoder blocks by index (reference models/vit.py:302-315)."""
        for i in sorted(remove_layers, reverse=True):
            del self.encoder.layers[i]


class VisionTransformer(_ViTBase):
    """ViT classifier (reference models/vit.py:100-248)."""

    def __init__(self, image_size: int, patch_size: int, num_layers: int, num_heads: int, hidden_dim: int,
                 mlp_dim: int, dropout: float = 0.0, attention_dropout: float = 0.0, num_classes: int = 1000,
                 representation_size: Optional[int] = None, num_registers: int = 0, num_class_tokens: int = 1,
                 torch_pretrained_weights: Optional[str] = None, timm_pretrained_weights: Optional[List] = None,
                 remove_layers: List[int] = []):
        super().__init__()
        seq_length = self._init_stem(image_size, patch_size, hidden_dim, mlp_dim, dropout, attention_dropout,
                                     num_classes, representation_size, num_heads, num_registers, num_class_tokens)
        if num_registers > 0:
            self.register_tokens = nn.Parameter(torch.zeros(1, num_registers, hidden_dim))
            seq_length += num_registers
        self.encoder = ViTEncoder(seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout, attention_dropout)
        self.seq_length = seq_length
        self._init_head()
        self.load_weights(torch_pretrained_weights, timm_pretrained_weights)
        if remove_layers:
            self.remove_layers(remove_layers)

    def forward(self, x: torch.Tensor):
        self._check_image(x)
        if x.shape[0] == 0:                    # a batch of zero images (the reference's nn.MultiheadAttention raises on it): empty logits
            return x.new_zeros((0, self.num_classes), dtype=torch.float32)
        if train_engine.train_eligible(x, self, max(self.dropout, self.attention_dropout)) and \
                train_engine.supported(self.hidden_dim, self.num_heads, self.seq_length):
            def body():
                tokens = train_engine.embed_tokens_train(self, x)  # same kernels, recorded for loss.backward()
                tokens = self.encoder(tokens, _pos_added=True, _rows=self.num_class_tokens)
                return train_engine.pool_and_head_train(self, tokens)
            with engine.on_device(x):
                return train_engine.model_forward_train(self, x, body)     # one training pass: operand type + loss scale (train_engine docstring)
        if engine.backend_for(x, self, max(self.dropout, self.attention_dropout)) == "hip":
            return engine.run_guarded(self, x, lambda: engine.forward_split(x, self._hip_forward), probe=self._hip_forward)
        return self._composite_head(self.encoder(self._composite_tokens(x)))

    def _hip_forward(self, x: torch.Tensor):
        tokens = engine.embed_tokens(self, x)                      # im2col + GEMM (+bias +pos), cls rows
        tokens = engine.call_module(self.encoder, tokens, _pos_added=True, _rows=self.num_class_tokens)      # blocks dispatch themselves
        return engine.pool_and_head(self, tokens)                  # LN on CLS rows, sum, fp32 head
