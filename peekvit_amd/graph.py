"""HIP-graph capture of the forward path for launch-bound (small batch) use.

A ViT-B/16 forward at batch 2048 spends ~85 ms in ~100 kernels, so launch cost is invisible there; at batch <= 64 or
for vit_tiny the ~100 ctypes launches (~5 us each on the host) dominate.  Every C-ABI entry point is stateless,
allocation-free and launches on the caller's stream (include/peekvit_hip.h), so the whole forward can be captured once
into a hipGraph and replayed ("HIP streams and graphs instead of a tracing compiler").
"""
from __future__ import annotations

import torch


class GraphedForward:
    """Capture `model(x)` for one input shape and replay it: `y = GraphedForward(model, example)(x)`.

    The model must be in eval mode on a GPU (the HIP path); weights are read through their bf16 cache, so call
    `refresh()` after changing parameters.  Input is copied into a static buffer, the output tensor is static
    (clone it if it must survive the next call)."""

    def __init__(self, model: torch.nn.Module, example: torch.Tensor, warmup: int = 2):
        assert example.is_cuda and not model.training, "GraphedForward needs an eval-mode model and a GPU tensor"
        self.model = model
        self.static_in = example.clone()
        self._capture(warmup)

    def _capture(self, warmup: int):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                 # populates the bf16 weight cache and the workspace arena OUTSIDE capture
                self.model(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self.model(self.static_in)

    def refresh(self):
        """Re-capture (after a parameter update or `set_budget`)."""
        self._capture(1)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
