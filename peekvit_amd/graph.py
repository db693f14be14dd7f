"""HIP-graph capture of the forward path for launch-bound (small batch) use.

A ViT-B/16 forward at batch 2048 spends ~85 ms in ~100 kernels, so launch cost is invisible there; at batch <= 64 or
for vit_tiny the ~100 ctypes launches (~5 us each on the host) dominate.  Every C-ABI entry point is stateless,
allocation-free and launches on the caller's stream (include/peekvit_hip.h), so the whole forward can be captured once
into a hipGraph and replayed ("HIP streams and graphs instead of a tracing compiler").
"""
from __future__ import annotations

import torch

from . import engine


class GraphedForward:
    """Capture `model(x)` for one input shape and replay it: `y = GraphedForward(model, example)(x)`.

    The model must be in eval mode on a GPU (the HIP path); weights are read through their bf16 cache, so call
    `refresh()` after changing parameters.  Input is copied into a static buffer, the output tensor is static
    (clone it if it must survive the next call).  The graph owns its scratch: warm-up and capture draw from a private workspace
    arena, so a later eager forward that grows the shared arena cannot free memory the captured nodes point at.  Precision mode
    "auto": the capture is the fp16-operand forward incl. the zeroing of the range flag; every replay reads the flag and a
    forward that tripped it is repeated eagerly (engine.run_guarded: folding off or the bf16x3 mode, whichever the flag asks for)."""

    def __init__(self, model: torch.nn.Module, example: torch.Tensor, warmup: int = 2, capture_error_mode: str = "global"):
        assert example.is_cuda and not model.training, "GraphedForward needs an eval-mode model and a GPU tensor"
        self.model = model
        self.static_in = example.clone()
        self._ws = engine._Workspace()
        self._flag = engine.range_flag_for(example.device)
        self._error_mode = capture_error_mode          # "thread_local" for engine-initiated captures (peekvit_amd.autograph): other threads keep launching
        self._capture(warmup)

    def _capture(self, warmup: int):
        # warm-up and capture run on ONE side stream (the workspace is keyed by stream): weights are cast and the private arena is
        # populated outside the capture, the captured launches then find every buffer in place
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.graph = torch.cuda.CUDAGraph()
        with engine.on_device(self.static_in), engine.use_workspace(self._ws), torch.no_grad():
            if self.static_in.dtype == torch.uint8:
                engine.norm_constants(self.model, self.static_in.device)       # (device constants of the split-operand mode's uint8 path: not creatable under capture)
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.model(self.static_in)
            # what gets captured: the guarded fp16 forward (its first node zeroes the flag word, the replayer reads it) - unless mode "auto"
            # has already sent this model to the fallback mode (parameter bounds, or three trips in a row during warm-up): that capture
            # neither zeroes nor writes the word, so the replayer must not read it (a stale bit would send every replay to eager)
            # (or the warm-up forwards' self-check measured this model / budget / batch size outside the contract on fp16 operands)
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=self._error_mode):
                self.static_out = self.model(self.static_in)
            self._guarded = engine._mode() == "auto" and engine.last_forward_guarded()
            gs = engine.guard_state(self.model)
            self._captured_state = (gs.unsafe, gs.no_fold, gs.hybrid, gs.mlp_hybrid)
        torch.cuda.current_stream().wait_stream(side)

    def refresh(self):
        """Re-capture (after a parameter update or `set_budget`)."""
        self._capture(1)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self.static_in.copy_(x)
        self.graph.replay()
        if self._guarded and int(self._flag.max().item()) != 0:      # a guard tripped inside the replay: this batch again, eagerly, in a mode that keeps the contract
            out = self.model(self.static_in).clone()
            gs = engine.guard_state(self.model)
            if (gs.unsafe, gs.no_fold, gs.hybrid, gs.mlp_hybrid) != self._captured_state:
                # the eager repeat changed what mode "auto" does for this model from now on (folding off, hybrid layers, or the split-operand mode for good):
                # capture THAT forward, or every later call would replay the graph that trips and then run eagerly again (round 3 ADVICE)
                self._capture(1)
            return out
        return self.static_out
