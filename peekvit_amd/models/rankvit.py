"""Token-ranking ViT behind the reference's interface (reference models/rankvit.py).

Blocks listed in `rankvit_layers` rank their non-class tokens by L2 norm and keep the top
ceil(N * budget) - in descending-norm order - before running the ordinary pre-LN block
(models/rankvit.py:55-97).  On the MI355X path the ranking is three HIP kernels (row norms, per-image
rank/top-k in LDS, coalesced row compaction); ties rank lowest-index-first (the reference's argsort is
unstable there).
"""
from __future__ import annotations

import math
from typing import List, Optional, Union

import torch
from torch import nn

from .. import engine, train_engine
from .vit import ViTBlock, _ViTBase, _make_layers


class RankViTBlock(ViTBlock):
    """ViTBlock preceded by sort_and_drop when its budget is not 1 (reference models/rankvit.py:24-101)."""

    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float, attention_dropout: float):
        super().__init__(num_heads, hidden_dim, mlp_dim, dropout, attention_dropout)
        self.sort = False
        self.current_budget = 1.0
        self.last_keep = None      # int32 [B,k] indices kept by the most recent HIP sort_and_drop (introspection/tests)

    def sort_and_drop(self, input: torch.Tensor):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        if train_engine.train_eligible(input, self, self._p_drop):
            out, self.last_keep = train_engine.sort_and_drop_train(input, self.current_budget)    # HIP ranking, scatter backward
            return out
        if engine.backend_for(input, self, self._p_drop) == "hip":
            with engine.on_device(input):
                out, self.last_keep = engine.sort_and_drop(input, self.current_budget)
            return out
        cls_tok, rest = input[:, :1], input[:, 1:]
        order = torch.argsort(torch.norm(rest, dim=-1), dim=-1, descending=True, stable=True)
        keep = order[:, :math.ceil(rest.shape[1] * self.current_budget)]
        self.last_keep = keep
        kept = torch.gather(rest, 1, keep.unsqueeze(-1).expand(-1, -1, rest.shape[-1]))
        return torch.cat([cls_tok, kept], dim=1)

    def forward(self, input: torch.Tensor):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        if self.current_budget != 1:
            input = self.sort_and_drop(input)
        return super().forward(input)

    def _pv_forward_rows(self, input: torch.Tensor, nq: int):
        """Last block of a model forward: rank / drop as forward() does, then only the class-token row of the block's output."""
        if type(self) is not RankViTBlock or nq != 1 or input.dim() != 3 or not engine.rows_only_ok(self):
            return None
        if not (train_engine.train_eligible(input, self, self._p_drop) or engine.backend_for(input, self, self._p_drop) == "hip"):
            return None
        if self.current_budget != 1:
            input = self.sort_and_drop(input)
        return self._pv_rows(input, nq)

    def set_budget(self, budget: float):
        self.current_budget = budget

    def _pv_plain_ln1(self) -> bool:
        return self.current_budget == 1      # with an active budget the tokens are ranked / dropped before ln_1

    def _pv_ranks_input(self) -> bool:
        return self.current_budget != 1      # engine hint: the producer of this block's input may leave the token norms behind


class RankViTEncoder(nn.Module):
    """Encoder whose blocks at `rankvit_layers` are RankViTBlocks (reference models/rankvit.py:105-152).
    Like the reference, `rankvit_layers=None` raises TypeError (`i in None`)."""

    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int, dropout: float,
                 attention_dropout: float, rankvit_layers: Optional[List[Union[int, float]]] = None):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.dropout = nn.Dropout(dropout)
        self.layers = _make_layers(
            lambda i: (RankViTBlock if i in rankvit_layers else ViTBlock)(num_heads, hidden_dim, mlp_dim, dropout,
                                                                         attention_dropout), num_layers)
        self.ln = nn.LayerNorm(hidden_dim)

    def forward(self, input: torch.Tensor, _pos_added: bool = False, _rows: int = 0):
        torch._assert(input.dim() == 3, f"Expected (batch_size, seq_length, hidden_dim) got {input.shape}")
        if _pos_added:
            return engine.run_layers(self.layers, input, last_rows=_rows)
        return self.ln(self.layers(self.dropout(input + self.pos_embedding)))


class RankVisionTransformer(_ViTBase):
    """reference models/rankvit.py:156-339."""

    def __init__(self, image_size: int, patch_size: int, num_layers: int, num_heads: int, hidden_dim: int,
                 mlp_dim: int, dropout: float = 0.0, attention_dropout: float = 0.0, num_classes: int = 1000,
                 representation_size: Optional[int] = None, num_registers: int = 0, num_class_tokens: int = 1,
                 torch_pretrained_weights: Optional[str] = None, timm_pretrained_weights: Optional[str] = None,
                 rankvit_layers: Optional[List[Union[int, float]]] = None):
        super().__init__()
        seq_length = self._init_stem(image_size, patch_size, hidden_dim, mlp_dim, dropout, attention_dropout,
                                     num_classes, representation_size, num_heads, num_registers, num_class_tokens)
        self.rankvit_layers = rankvit_layers
        if num_registers > 0:
            raise ValueError("Registers are not supported yet for this model.")
        self.encoder = RankViTEncoder(seq_length, num_layers, num_heads, hidden_dim, mlp_dim, dropout,
                                      attention_dropout, rankvit_layers)
        self.seq_length = seq_length
        self._init_head()
        self.load_weights(torch_pretrained_weights, timm_pretrained_weights)

    def forward(self, x: torch.Tensor):
        self._check_image(x)
        if x.shape[0] == 0:                    # a batch of zero images (the reference's nn.MultiheadAttention raises on it): empty logits
            return x.new_zeros((0, self.num_classes), dtype=torch.float32)
        if train_engine.train_eligible(x, self, max(self.dropout, self.attention_dropout)) and \
                train_engine.supported(self.hidden_dim, self.num_heads, self.seq_length):
            def train_body():
                tokens = self.encoder(train_engine.embed_tokens_train(self, x), _pos_added=True, _rows=self.num_class_tokens)
                return train_engine.pool_and_head_train(self, tokens)
            with engine.on_device(x):
                return train_engine.model_forward_train(self, x, train_body)
        if engine.backend_for(x, self, max(self.dropout, self.attention_dropout)) == "hip":
            plain = lambda xs: engine.pool_and_head(self, engine.call_module(self.encoder, engine.embed_tokens(self, xs), _pos_added=True, _rows=self.num_class_tokens))
            # what a ranked layer decided for every image: its kept SET (sorted indices); mode auto's self-check compares arithmetic only where
            # these agree with the split-operand run (engine.RANK_STRICT)
            ranked = [blk for blk in self.encoder.layers if getattr(blk, "current_budget", 1) != 1 and hasattr(blk, "sort_and_drop")]
            state = (lambda: [torch.sort(blk.last_keep, dim=1).values for blk in ranked if getattr(blk, "last_keep", None) is not None]) if ranked else None
            body = (lambda xs: self._forward_repairing_near_ties(xs, plain, ranked)) if ranked and engine.RANK_REPAIR else plain
            object.__setattr__(self, "_pv_no_autograph", bool(ranked and engine.RANK_REPAIR))      # (the repair reads a count on the host: not a forward to capture)
            return engine.run_guarded(self, x, lambda: body(x), probe=body, probe_key=repr(getattr(self, "current_budget", None)), probe_state=state)
        return self._composite_head(self.encoder(self._composite_tokens(x)))

    def _forward_repairing_near_ties(self, x: torch.Tensor, plain, ranked):
        """The forward with its keep boundaries watched (engine.RANK_REPAIR, round 6): on 16-bit operands every ranking also reports each image's
        relative gap between its last kept and its first dropped token norm; the images whose narrowest gap is under engine.RANK_TIE_GAP - where the
        16-bit layers' noise in the norms (~1e-4) may have resolved a near-tie differently from the reference's fp32 arithmetic (models/rankvit.py:63-77)
        - are run AGAIN as a small sub-batch in the split-operand arithmetic (kept sets bit-exact end to end), their logits and the blocks' `last_keep`
        rows replaced.  Everything else stays on fp16 operands.  One extra host read per forward (the count of such images)."""
        if engine._mode() != "f16" or torch.cuda.is_current_stream_capturing():       # the split-operand arithmetic itself / an unguarded A/B mode / a capture: nothing to repair
            return plain(x)
        with engine.rank_gaps(int(x.shape[0]), x.device) as gap:
            out = plain(x)
        engine.rank_repair_forwards += 1
        idx = torch.nonzero(gap < engine.RANK_TIE_GAP).flatten()                       # (host synchronisation: the size of idx)
        if idx.numel() == 0:
            return out
        if 2 * int(idx.numel()) > int(x.shape[0]):
            # a dense boundary (random weights on random images: 87 % of the images sit within 4e-4 of a tie, profiles/r06_rank_tie_calibration.json):
            # the whole batch in split precision, no gather
            engine.rank_repaired_images += int(x.shape[0])
            with engine.precision(engine.FALLBACK_MODE), engine._hooks_held(self):      # (module hooks have seen this batch once already)
                return plain(x)
        keeps = [blk.last_keep for blk in ranked]
        with engine.precision(engine.FALLBACK_MODE), engine._hooks_held(self):
            fixed = plain(x.index_select(0, idx))
        out = out.clone() if out.is_inference() else out
        out.index_copy_(0, idx, fixed.to(out.dtype))
        for blk, kp in zip(ranked, keeps):                                             # what the blocks remember is the whole batch's, repaired rows included
            if kp is not None and getattr(blk, "last_keep", None) is not None and blk.last_keep.shape[1:] == kp.shape[1:]:
                kp = kp.clone() if kp.is_inference() else kp
                kp.index_copy_(0, idx, blk.last_keep.to(kp.dtype))
                blk.last_keep = kp
        engine.rank_repaired_images += int(idx.numel())
        return out

    def set_budget(self, budget: float):
        """Only the blocks in rankvit_layers receive it; a list is indexed by LAYER index
        (reference models/rankvit.py:283-288)."""
        self.current_budget = budget
        for i in self.rankvit_layers:
            self.encoder.layers[i].set_budget(budget[i] if isinstance(budget, list) else budget)
