"""Training path of the MI355X engine: forward WITH saved activations and the hand-written backward, exposed to PyTorch as
autograd Functions so that the reference's loop (train/train.py:112-121: `out = model(x)`, `loss.backward()`,
`optimizer.step()`) runs unchanged.  Pure plumbing like engine.py: every product / reduction is a C-ABI kernel.

Per encoder block (reference models/vit.py:45-55), R = B*S token rows:
  forward   h1 = LN1(x) | qkv = h1.Win^T+b (q pre-scaled) | att = attention(qkv) | x1 = x + att.Wo^T+b | h2 = LN2(x1)
            [gl | dgl] = fc1 epilogue pair: pre = h2.W1^T+b, gl = gelu(pre), dgl = gelu'(pre) | out = x1 + gl.W2^T+b
                                                                         saved: x, h1, qkv, att, x1, h2, [gl | dgl]
  backward  d2 = bf16(dout)
            dpre = (d2.W2)*dgl         [one GEMM, PV_EPI_GELU_GRAD_BF16]      dW2 = d2^T.gl     db2 = colsum(d2)
            dh2 = dpre.W1      dW1 = dpre^T.h2   db1 = colsum(dpre)
            dx1 = dout + LN2'(dh2)               (dgamma2, dbeta2)
            d1 = bf16(dx1)
            datt = d1.Wo       dWo = d1^T.att    dbo = colsum(d1)
            dqkv = attention'(qkv, datt)
            dh1 = dqkv.Win     dWin = dqkv^T.h1  dbin = colsum(dqkv)
            dx = dx1 + LN1'(dh1)                 (dgamma1, dbeta1)
Data gradients (x.W) are the forward NT GEMM on the transposed bf16 weight; weight gradients (dY^T.X) are the same NT GEMM on
the transposed activations with split-K over the R rows (pv_transpose_bf16 + pv_gemm_bf16 ksplit + pv_sum_slices_f32).
Gradients of 16-bit tensors travel in 16 bits (as under torch autocast); the residual-stream gradient and all parameter
gradients are fp32.

Operand type (round 5).  A MODEL-level training forward in precision mode "auto" runs on IEEE fp16 operands - the library the inference
path uses - so that the training forward sits inside the same 1e-3 contract as evaluation (bf16 operands: 4.7e-3 at ViT-B/16).  fp16
gradients need a LOSS SCALE: everything between the head and the stem - the HIP chain - runs in "scaled space", L * dL/dx, with L a power
of two chosen so that the largest gradient entering the chain sits at 2^7: measured on the golden models (scripts/train_f16_probe.py,
profiles/r05_train_f16_probe.txt) the largest 16-bit gradient inside the chain is 12x the entering one (the class-row gradient fans out
through LN2' and the attention), so the largest value anywhere sits near 1.5e3 - 40x below 65504 - and the smallest per-tensor maximum near 2,
whose typical elements are still 300x above fp16's smallest normal number (the fp16 MFMA flushes subnormal operands).  The chain's boundaries multiply by L (where a gradient enters: behind
the last block, `block.mask` handed to auxiliary losses) or by 1/L (where one leaves: every parameter gradient, the stem, the budget
token) - see enter() / leave() / _unscale() - so autograd and the optimizer only ever see true gradients: the reference's loop
(train/train.py:112-121) runs unchanged.  L follows the entering gradient's measured maximum with one step's delay (the first step reads
it synchronously).  An fp16 overflow anywhere in the chain turns into inf / NaN that propagates to the small per-block reductions every
backward node adds to `TrainPass.chk`.  The verdict is formed ON THE DEVICE at the end of the backward pass (no host synchronisation: the host
keeps queueing the next step) and applied where the step would be taken, by a global optimizer-step pre-hook, the way torch.cuda.amp.GradScaler
does: a FUSED torch optimizer receives it as `found_inf` (its kernel skips the update), any other optimizer has every parameter's `.grad` set to
None behind one event wait, which makes `step()` a no-op; the scale target is lowered when the host gets to read the verdict (at the latest when
the next backward pass starts).  An overflow of the FORWARD (a 16-bit activation beyond 65504: range flag bit 1) sends that model's training to
bf16 operands for good.
Blocks called on their own under autograd (no model-level pass) keep bf16 operands and no scale, as in rounds 1-4.
"""
from __future__ import annotations

import contextlib
import math
import os
import threading
import warnings
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops
from ._lib import (PV_EPI_BIAS_BF16, PV_EPI_BIAS_F32, PV_EPI_BIAS_GELU_PAIR_BF16, PV_EPI_BIAS_RES_F32, PV_EPI_GELU_GRAD_BF16)
from .engine import _f32, bf16_weight, pver, workspace

# ------------------------------------------------------------------------------------------------------------------------------------
# operand type + loss scale of one training pass (module docstring)
# ------------------------------------------------------------------------------------------------------------------------------------
# PEEKVIT_AMD_TRAIN_OPERAND: "" (default) = fp16 in precision mode "auto", bf16 in mode "bf16"; "bf16" / "f16" force one
_TRAIN_OPERAND = os.environ.get("PEEKVIT_AMD_TRAIN_OPERAND", "")
SCALE_TARGET = float(os.environ.get("PEEKVIT_AMD_TRAIN_SCALE_TARGET", "128"))    # L * max|entering gradient|
# An overflow divides the target by 4; round 5 never raised it again, so a few transient spikes left L 64x lower for the rest of a run - the smallest
# per-tensor gradient maxima then sit in fp16's flush-to-zero range: silent precision loss instead of a skip.  Like GradScaler's
# growth_interval: after this many CONSECUTIVE clean steps the target doubles, up to SCALE_TARGET.
SCALE_GROWTH_INTERVAL = int(os.environ.get("PEEKVIT_AMD_TRAIN_SCALE_GROWTH_INTERVAL", "200"))
_tls = threading.local()
debug_amax = None            # a list: BlockFn.backward appends the maxima of its 16-bit gradients (diagnostics only)
steps_skipped = 0            # training steps skipped because an fp16 gradient overflowed (tests / bench read it)
forward_fallbacks = 0        # models sent to bf16-operand training because their fp16 forward overflowed


class TrainState:
    """What the training path has learnt about one model (plain attribute `_pv_train`)."""
    __slots__ = ("operand", "target", "amax", "scale", "steps", "skipped", "last_skipped", "on_skip", "warned", "pending", "clean")

    def __init__(self):
        self.operand = None            # None: decided per pass from the precision mode; "bf16": sticky (an fp16 forward overflowed / weights do not fit)
        self.target = SCALE_TARGET
        self.amax = None               # max |gradient entering the chain| of the last finished pass (host float)
        self.scale = 1.0               # L of the last pass
        self.steps, self.skipped, self.last_skipped = 0, 0, False
        self.clean = 0                 # consecutive clean steps since the target was last lowered (the target grows back after SCALE_GROWTH_INTERVAL of them)
        self.on_skip = None            # callable(model) -> None replacing the default "every .grad = None" (dist.OverlappedGradReducer)
        self.warned = False
        self.pending = None            # the last backward pass whose verdict has not been read on the host yet


def train_state(model: nn.Module) -> TrainState:
    st = getattr(model, "_pv_train", None)
    if st is None:
        st = TrainState()
        object.__setattr__(model, "_pv_train", st)
    return st


def last_step_skipped(model: nn.Module) -> bool:
    """Did the last backward through `model` overflow fp16?  Waits for that pass's verdict (one event) and, if it is one to skip and no optimizer
    step has taken care of it yet, drops its gradients (every .grad = None): call it right behind `loss.backward()`."""
    st = train_state(model)
    tp = st.pending
    if tp is not None:
        if tp.resolve() and not tp.applied:
            tp.drop_gradients()
            if tp in _pending_passes:
                _pending_passes.remove(tp)
    return st.last_skipped


class TrainPass:
    """One model-level training forward and the backward passes through its graph."""

    def __init__(self, model: Optional[nn.Module], state: Optional[TrainState], operand: str, device):
        self.model, self.state, self.operand = model, state, operand
        self.scaled = operand == "f16" and model is not None
        self.L = 1.0
        self.active = False
        with torch.inference_mode(False), torch.no_grad():
            self.flag = torch.zeros(1, dtype=torch.int32, device=device) if self.scaled else None
            self.chk = torch.zeros(1, dtype=torch.float32, device=device) if self.scaled else None
        self.fwd_flag = None
        self.amax = None
        self.found, self.host, self.event, self.applied, self.skipped = None, None, None, True, False
        self.stepped = set()           # ids of the model's parameters an optimizer has already stepped (or skipped) under this pass's verdict
        self._pids = None

    # -- backward side ---------------------------------------------------------------------------------------------------------
    def begin_backward(self, grad: torch.Tensor, primary: bool) -> float:
        """Called by the boundary nodes: the first one of a backward pass fixes L and queues the end-of-pass callback."""
        if not self.active:
            self.active = True
            st = self.state
            if st.pending is not None and st.pending is not self:
                st.pending.resolve()           # the previous step's verdict and statistics (its event has long fired: the GPU is at least a forward further)
            amax = st.amax
            if amax is None and primary:
                amax = float(grad.detach().abs().max())            # first pass of this model: one synchronous read
            if amax is None or not math.isfinite(amax) or amax <= 0.0:
                self.L = 1.0
            else:
                self.L = 2.0 ** max(min(math.floor(math.log2(st.target / amax)), 40), -40)
            st.scale = self.L
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        if primary:
            with torch.no_grad():
                self.amax = grad.detach().abs().max()
        return self.L

    def note(self, t: torch.Tensor):
        """A small fp32 reduction of this node's gradients (LayerNorm-backward column sums, ...): inf / NaN anywhere upstream shows here."""
        if self.scaled:
            with torch.no_grad():
                self.chk.add_(t.sum())

    def _finish(self):
        """End of a backward pass (autograd callback): NO host synchronisation here (round 5, second form - the first one read the check word on
        the spot, which cost small models a third of their training throughput: the host could no longer queue the next step while this one
        runs).  The verdict is formed ON THE DEVICE - `found` = 1.0 if the check word is not finite or the forward raised its overflow bit - and
        copied with the statistics to pinned host memory behind the kernels; who needs it on the host waits for that one event (`resolve`).
        It is APPLIED where the step would otherwise be taken: the global optimizer-step pre-hook below."""
        self.active = False
        with torch.no_grad():
            fwd = (self.fwd_flag if self.fwd_flag is not None else self.flag)
            self.found = torch.logical_or(~torch.isfinite(self.chk), (fwd & 1) != 0).float()          # [1] on the device
            words = torch.cat([self.chk, (self.amax if self.amax is not None else self.chk.new_zeros(())).reshape(1), fwd.float(), self.found])
            self.found = self.found.reshape(())            # (torch's fused optimizers take `found_inf` as a 0-dim tensor, like GradScaler's)
            self.host = torch.empty(4, dtype=torch.float32).pin_memory()
            self.host.copy_(words, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(self.chk.device))
            self.chk = torch.zeros_like(self.chk)          # (a later backward pass through the same graph starts a fresh word)
        self.amax = None
        st = self.state
        if st.pending is not None and st.pending is not self:
            st.pending.resolve()
        st.pending = self
        self.applied = False
        # (a pass some optimizer has already consumed is over once the model's NEXT backward ends: parameters no optimizer owns never complete it)
        _pending_passes[:] = [tp for tp in _pending_passes if not (tp.state is st and tp.stepped)]
        _pending_passes.append(self)
        while len(_pending_passes) > 16:           # (backward passes that no optimizer step ever consumes - gradients taken for their own sake - must not pile up)
            old = _pending_passes.pop(0)
            old.resolve()
            old.applied = True

    def resolve(self) -> bool:
        """Wait for this pass's verdict (one event) and book it: loss-scale statistics, skipped-step counters, the bf16 fallback after a forward
        overflow.  Returns whether the step was one to skip.  Idempotent."""
        global steps_skipped, forward_fallbacks
        if self.host is None:
            return self.skipped
        self.event.synchronize()
        chk, amax, fbits, found = self.host.tolist()
        self.host = None
        st = self.state
        if st.pending is self:
            st.pending = None
        fwd_over = (int(fbits) & 1) != 0
        self.skipped = bool(found)
        if not self.skipped:
            st.steps += 1
            st.last_skipped = False
            if amax > 0.0 and math.isfinite(amax):
                st.amax = amax
            if st.target < SCALE_TARGET and SCALE_GROWTH_INTERVAL > 0:
                st.clean += 1
                if st.clean >= SCALE_GROWTH_INTERVAL:
                    st.target, st.clean = min(st.target * 2.0, SCALE_TARGET), 0
            return False
        st.skipped += 1
        st.last_skipped = True
        st.clean = 0
        steps_skipped += 1
        if fwd_over:
            st.operand = "bf16"
            forward_fallbacks += 1
            warnings.warn("peekvit_amd: a 16-bit activation of the training forward left the fp16 range (|v| > 65504); this step's gradients were "
                          "dropped and this model trains on bf16 operands from now on", RuntimeWarning, stacklevel=2)
        else:
            st.target = max(st.target / 4.0, 2.0)
            st.amax = amax if amax > 0.0 and math.isfinite(amax) else st.amax
            if not st.warned:
                st.warned = True
                warnings.warn(f"peekvit_amd: an fp16 gradient overflowed at loss scale {self.L:g}; this step's gradients were dropped (the optimizer "
                              "step is skipped) and the scale target was lowered", RuntimeWarning, stacklevel=2)
        return True

    def param_ids(self):
        if self._pids is None:
            self._pids = frozenset(id(p) for p in self.model.parameters() if p.requires_grad)
        return self._pids

    def drop_gradients(self):
        """The skip, host form: every .grad = None (torch optimizers and clip_grad_norm_ pass over such parameters), or the owner's own way
        (dist.OverlappedGradReducer: its gradients alias the all-reduce buckets)."""
        if self.applied:
            return
        self.applied = True
        if self.state.on_skip is not None:
            self.state.on_skip(self.model)
        else:
            for p in self.model.parameters():
                p.grad = None


# Passes whose verdict has not been applied to an optimizer step yet.  The global optimizer-step pre-hook applies them: a FUSED torch optimizer
# (`_step_supports_amp_scaling`: the interface torch.cuda.amp.GradScaler uses) is handed the device-side verdict as `found_inf` - its kernel skips
# the update and takes the step count back, no host synchronisation; any other optimizer gets the host form: wait for the verdict, and if the
# step is one to skip set every .grad to None, which makes `step()` a no-op (what GradScaler does for such optimizers, synchronisation included).
_pending_passes: list = []


def book_external_skip(model: nn.Module):
    """A step of `model` is being skipped for a reason its own backward pass did not see (non-finite values in all-reduced gradients)."""
    global steps_skipped
    st = train_state(model)
    st.skipped += 1
    st.last_skipped = True
    st.clean = 0
    steps_skipped += 1


def _optimizer_pre_hook(opt, args, kwargs):
    if not _pending_passes:
        return None
    mine = {id(p) for g in opt.param_groups for p in g["params"]}
    # Round 6: a pass stays pending until EVERY parameter of its model has been stepped under its verdict (or the model's next backward
    # ends).  Round 5 let the first optimizer that shared a parameter with the model consume the pass: a second optimizer over the model's other
    # parameters (backbone / head, gates split off) then found nothing pending and stepped on the overflowed gradients.
    todo = [tp for tp in _pending_passes if (tp.param_ids() & mine) - tp.stepped]
    if not todo:
        return None
    for tp in todo:
        tp.stepped |= tp.param_ids() & mine
        if tp.stepped >= tp.param_ids():
            _pending_passes.remove(tp)
    # The verdict of a pass covers ITS backward.  Whoever sums gradients across ranks afterwards (torch's DistributedDataParallel, a hand-written
    # all-reduce) hands a rank whose own verdict says "clean" the inf / NaN of a rank that overflowed: the gradients this step is about to consume
    # are therefore checked as well (one multi-tensor norm over them - the pass clip_grad_norm_ makes too), on the device for fused optimizers.
    grads = [p.grad for g in opt.param_groups for p in g["params"] if p.grad is not None and p.grad.is_cuda]
    if getattr(opt, "_step_supports_amp_scaling", False):
        # fused optimizer: the device-side verdict travels as `found_inf` - also when the gradients alias an OverlappedGradReducer's buckets
        # (round 5 sent that case to the host form below, whose "every .grad = None" the reducer turns into a no-op: the step then ran)
        found = todo[0].found if len(todo) == 1 else torch.stack([tp.found for tp in todo]).amax(0)
        if grads:
            with torch.no_grad():
                found = torch.maximum(found, (~torch.isfinite(torch.stack(torch._foreach_norm(grads)).sum())).float().to(found.device))
        prev = getattr(opt, "found_inf", None)
        opt._pv_prev_found = prev
        opt.found_inf = found if prev is None else torch.maximum(prev.to(found.device).reshape(()).float(), found)
        opt._pv_found_set = True
        for tp in todo:
            tp.applied = True
        return None
    skipped = False
    for tp in todo:
        if tp.resolve():
            skipped = True
            if tp.state.on_skip is not None and not tp.applied:
                raise RuntimeError("peekvit_amd: this training step overflowed fp16 and must be skipped, but its gradients alias an OverlappedGradReducer's "
                                   "buckets and cannot be dropped for a non-fused optimizer: call optimizer.step() only `if not reducer.skip_step` "
                                   "(or use a fused torch optimizer, which receives the verdict as found_inf)")
            tp.applied = False
            tp.drop_gradients()
        tp.applied = True
    if not skipped and grads:
        with torch.no_grad():
            if not bool(torch.isfinite(torch.stack(torch._foreach_norm(grads)).sum())):       # (non-finite gradients from somewhere else: see above)
                if any(tp.state.on_skip is not None for tp in todo):
                    raise RuntimeError("peekvit_amd: non-finite gradients in an OverlappedGradReducer's buckets at optimizer.step(): check "
                                       "`reducer.skip_step` after finish() (it covers the reduced buckets) or use a fused torch optimizer")
                for tp in todo:
                    book_external_skip(tp.model)
                    tp.applied = False
                    tp.drop_gradients()
    return None


def _optimizer_post_hook(opt, args, kwargs):
    if getattr(opt, "_pv_found_set", False):
        opt.found_inf = opt._pv_prev_found
        opt._pv_found_set = False


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post, register_optimizer_step_pre_hook as _reg_pre
    _reg_pre(_optimizer_pre_hook)
    _reg_post(_optimizer_post_hook)
except ImportError:                    # pragma: no cover - torch < 2.0
    pass


def current_pass() -> Optional[TrainPass]:
    return getattr(_tls, "tp", None)


def pass_operand(model: nn.Module) -> str:
    """The operand type a model-level training forward of `model` uses now."""
    from . import engine
    st = train_state(model)
    if st.operand is not None:
        return st.operand
    if _TRAIN_OPERAND in ("bf16", "f16"):
        return _TRAIN_OPERAND
    return "f16" if engine._PRECISION == "auto" else "bf16"


@contextlib.contextmanager
def _kernels(tp: Optional[TrainPass], operand: Optional[str] = None):
    """Operand library + range flag of a pass on the CALLING thread (the backward runs on autograd's device thread)."""
    op = tp.operand if tp is not None else operand
    if op is None:
        yield
        return
    from . import engine
    old_op = _lib.set_operand(op)
    old_flag = ops.current_range_flag()
    ops.set_range_flag(tp.flag if tp is not None and tp.scaled else None)
    try:
        if tp is not None and tp.scaled:
            with engine.no_param_checks():
                yield
        else:
            yield
    finally:
        _lib.set_operand(old_op)
        ops.set_range_flag(old_flag)


def model_forward_train(model: nn.Module, x: torch.Tensor, body):
    """Run `body()` (embed -> encoder -> pool_and_head_train, recorded by autograd) as one training pass of `model`."""
    from . import engine
    st = train_state(model)
    for _attempt in range(2):
        tp = TrainPass(model, st, pass_operand(model), x.device)
        old = current_pass()
        _tls.tp = tp
        try:
            with _kernels(tp):
                out = body()
            if tp.scaled:
                with torch.no_grad():
                    tp.fwd_flag = tp.flag.clone()
            else:
                st.last_skipped = False          # (a pass without a loss scale has nothing to overflow: bf16 has fp32's range)
            return out
        except engine.F16RangeError as e:
            if tp.operand != "f16":
                raise
            st.operand = "bf16"
            warnings.warn(f"peekvit_amd: {e}; this model trains on bf16 operands", RuntimeWarning, stacklevel=2)
        finally:
            _tls.tp = old
    raise AssertionError("unreachable")


class _ScaleGradFn(torch.autograd.Function):
    """Identity whose backward multiplies by L (a gradient ENTERING the scaled chain) or by 1 / L (one LEAVING it)."""

    @staticmethod
    def forward(ctx, x, tp, enter, primary):
        ctx.tp, ctx.enter, ctx.primary = tp, enter, primary
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        tp = ctx.tp
        if ctx.enter:
            L = tp.begin_backward(g, ctx.primary)
            return (g if L == 1.0 else g * L), None, None, None
        L = tp.L
        if ctx.primary:                   # the gradient that leaves through the stem's input: the chain's last word
            tp.note(g.reshape(g.shape[0], -1).sum(0))
        return (g if L == 1.0 else g * (1.0 / L)), None, None, None


def enter(t: torch.Tensor, primary: bool = False, tp: Optional[TrainPass] = None) -> torch.Tensor:
    """Mark `t` as the place where gradients ENTER the scaled HIP chain from stock autograd (forward: identity)."""
    tp = current_pass() if tp is None else tp
    return _ScaleGradFn.apply(t, tp, True, primary) if tp is not None and tp.scaled and t.requires_grad else t


def leave(t: torch.Tensor, primary: bool = False, tp: Optional[TrainPass] = None) -> torch.Tensor:
    """Mark `t` as a place where gradients LEAVE the scaled chain into stock autograd (a stock-op parameter path feeding the chain)."""
    tp = current_pass() if tp is None else tp
    return _ScaleGradFn.apply(t, tp, False, primary) if tp is not None and tp.scaled and t.requires_grad else t


@contextlib.contextmanager
def stock_region():
    """A stretch of stock autograd ops INSIDE a model-level pass (between leave() and enter()): HIP functions called in there are on their
    own - no pass, bf16 operands, true gradients - exactly as when a block is called outside any model forward."""
    old_tp = current_pass()
    _tls.tp = None
    old_op = _lib.set_operand("bf16")
    old_flag = ops.current_range_flag()
    ops.set_range_flag(None)
    try:
        yield
    finally:
        _tls.tp = old_tp
        _lib.set_operand(old_op)
        ops.set_range_flag(old_flag)


def _unscale(tp: Optional[TrainPass], *grads):
    """Parameter gradients leave the chain: one multi-tensor multiply by 1 / L (in place)."""
    if tp is None or not tp.scaled or tp.L == 1.0:
        return
    ts, seen = [], set()
    for g in grads:
        if g is not None and g.data_ptr() not in seen:
            seen.add(g.data_ptr())
            ts.append(g)
    if ts:
        torch._foreach_mul_(ts, 1.0 / tp.L)


_wtcache: Dict[tuple, Tuple["weakref.ref", int, int, torch.Tensor]] = {}


def bf16_weight_t(p: torch.Tensor) -> torch.Tensor:
    """16-bit TRANSPOSE [K, N] of an fp32 [N, K] parameter (the 'weight' of the data-gradient GEMM), cached per version and operand type."""
    key = (id(p), _lib.OPERAND)
    ent = _wtcache.get(key)
    if ent is not None and ent[0]() is p and ent[1] == pver(p) and ent[2] == p.data_ptr():
        return ent[3]
    with torch.inference_mode(False):
        wt = ops.transpose(bf16_weight(p))
    _wtcache[key] = (weakref.ref(p, lambda _r, k=key: _wtcache.pop(k, None)), pver(p), p.data_ptr(), wt)
    return wt


def supported(D: int, H: int, S: int) -> bool:
    """Shapes the backward kernels cover (include/peekvit_hip.h): dh in {32,48,64}, S <= 208 (416 at dh = 32), D <= 1024."""
    dh = D // H
    return D % H == 0 and dh in (32, 48, 64) and S <= (416 if dh == 32 else 208) and D <= 1024 and D % 8 == 0


def _pad_rows(R: int) -> int:
    """K of the weight-gradient GEMMs: R rounded up so that up to 32 (big batches) / 8 split-K slices stay multiples of 128."""
    q = 4096 if R >= 65536 else 1024
    return (R + q - 1) // q * q


def _wgrad_transposed(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor, accumulate: bool, db=None):
    """out (+)= dy^T . x through transposed copies (any shape): pv_transpose_bf16 x 2 -> NT GEMM with split-K -> slice sum.
    db (optional fp32 [No]): column sums of dy from the same transposition pass."""
    R, No = dy.shape
    Ni = x.shape[1]
    Rp = _pad_rows(R)
    dev = dy.device
    dy_t = ops.transpose(dy, workspace.get("wg_a", (No, Rp), _lib.operand_dtype(), dev), pad_to=Rp, colsum_out=db)
    x_t = ops.transpose(x, workspace.get("wg_b", (Ni, Rp), _lib.operand_dtype(), dev), pad_to=Rp)
    tiles = ((No + 255) // 256) * ((Ni + 255) // 256)
    ksplit = 1
    while ksplit < 32 and tiles * ksplit < 512 and Rp % (ksplit * 2 * 128) == 0:
        ksplit *= 2
    part = workspace.get("wg_part", (ksplit, No, Ni), torch.float32, dev)
    ops.gemm(dy_t, x_t, None, part, PV_EPI_BIAS_F32, ksplit=ksplit, tag="[wgrad]")
    return ops.sum_slices(part, out, accumulate)


_ATTN_BWD = os.environ.get("PEEKVIT_AMD_ATTN_BWD", "lse")          # "lse": the persistent kernel from the forward's row statistics where it applies | "recompute"


def _attn_lse(B: int, S: int, H: int, dh: int, dev):
    """fp32 [B, H, S] for the forward's log-sum-exp rows when the persistent backward kernel serves the shape (ops.attention_bwd_lse_ok), else None."""
    if _ATTN_BWD != "lse" or not ops.attention_bwd_lse_ok(S, dh):
        return None
    return torch.empty((B, H, S), dtype=torch.float32, device=dev)


_SPLITK_MODE = os.environ.get("PEEKVIT_AMD_WGRAD_SPLIT", "balanced")      # "balanced" (r2) | "pow2" (r1)


def _tn_slices(R: int, tiles: int, No: int, Ni: int, cus: int = 256) -> int:
    """Number of split-K slices of the weight-gradient GEMM.  A launch is tiles x slices workgroups of equal length on `cus` CUs, so its
    time is ROUNDS x one workgroup: r1 took the first power of two with >= 512 workgroups (fc1 / fc2: 36 tiles x 16 = 576 = 2.25 rounds -> 3
    rounds at 75 %, out-proj: 9 x 32 = 288 = 1.125 -> 2 rounds at 56 %).  Now: the slice count (<= 32, whole 128-row blocks, the kernel
    gives the last slice the remainder) that minimises rounds x rows-per-slice + the fp32 partial-slab traffic it causes."""
    best, best_t = 1, float("inf")
    for s_ in range(1, 33):
        if R // (s_ * 2) < 256 and s_ > 1:
            break
        rounds = -(-(tiles * s_) // cus)
        t_gemm = rounds * (R / s_) * (2.0 * 256 * 256) / 4.1e12          # one 256^2 tile-row at ~1.05 PFLOP/s / 256 CUs
        t_part = s_ * No * Ni * 8.0 / 5.5e12                              # slabs written once, read once by pv_sum_slices_f32
        if t_gemm + t_part < best_t:
            best, best_t = s_, t_gemm + t_part
    return best


def _wgrad(dy: torch.Tensor, x: torch.Tensor, tag: str, bias_grad: bool = True):
    """(dW fp32 [No, Ni] = dy^T . x,  db fp32 [No] = column sums of dy) for bf16 dy [R, No], x [R, Ni] (row-strided views
    allowed).  128-multiples of features: the TN kernel reads the row-major activations directly, split-K over the rows; the
    < 128 * ksplit tail rows (and every other shape) go through the transposed-copy path."""
    R, No = dy.shape
    Ni = x.shape[1]
    dev = dy.device
    out = torch.empty((No, Ni), dtype=torch.float32, device=dev)
    db = torch.empty((No,), dtype=torch.float32, device=dev) if bias_grad else None
    if No % 128 == 0 and Ni % 128 == 0 and R >= 256 and not _NO_TN:
        tiles = ((No + 255) // 256) * ((Ni + 255) // 256)
        if _SPLITK_MODE == "pow2":
            s_ = 1
            while s_ < 32 and tiles * s_ < 512 and R // (s_ * 2) >= 256:
                s_ *= 2
        else:
            s_ = _tn_slices(R, tiles, No, Ni)
        r_main = R // 128 * 128                      # the TN kernel deals whole 128-row blocks to the slices (last slice: the rest)
        part = workspace.get("wg_part", (s_, No, Ni), torch.float32, dev)
        ops.gemm_tn(dy[:r_main], x[:r_main], part, s_)
        ops.sum_slices(part, out)
        if r_main < R:
            _wgrad_transposed(dy[r_main:], x[r_main:], out, True)
        if bias_grad:
            ops.colsum(dy, db)
        return out, db
    _wgrad_transposed(dy, x, out, False, db)
    return out, db


_NO_TN = os.environ.get("PEEKVIT_AMD_WGRAD", "tn") != "tn"
# The residual gradient at x1 (between the two LayerNorm backward kernels of a block) travels in 16 bits in an fp16 pass: LayerNorm 2's backward
# writes only the 16-bit copy the data / weight-gradient GEMMs read anyway, LayerNorm 1's backward reads that copy as its residual term - 26
# instead of 32 bytes per element through the two kernels that sit at the HBM roofline: 235.2 -> 232.0 ms per ViT-B/16 step (8.71 -> 8.83 k img/s).
# Cost: one fp16 rounding (2^-11, in scaled space) of the residual gradient per block - complete gradients 0.88 - 1.21e-3 from the reference's training
# step instead of 0.36 - 0.98e-3 (asserted 2e-3), gradient norms unchanged (2 - 4e-4).  The block-to-block stream stays fp32.  bf16 passes keep the
# fp32 hand-off (a bf16 rounding per block is 1 %).  PEEKVIT_AMD_TRAIN_DX1=f32 restores it everywhere.
_DX1_16 = os.environ.get("PEEKVIT_AMD_TRAIN_DX1", "16") == "16"


def _bf16_grad(dout: torch.Tensor, buf: torch.Tensor):
    """(bf16 copy of an incoming fp32 gradient, its column sums or None): the producer's LayerNorm-backward kernel already wrote one (attached to the
    very tensor object autograd hands over) unless the gradient comes from stock ops (the head) or was touched since."""
    hand = getattr(dout, "_pv_bf16", None)
    if hand is not None and hand[1] == dout._version and hand[0].numel() == buf.numel():
        return hand[0].view(buf.shape), hand[2]
    return ops.cast_bf16((dout if dout.is_contiguous() else dout.contiguous()).view(buf.shape), buf), None


class BlockFn(torch.autograd.Function):
    """One pre-LN encoder block with the MI355X forward + backward."""

    @staticmethod
    def forward(ctx, blk, x, ln1w, ln1b, inw, inb, ow, ob, ln2w, ln2b, w1, b1, w2, b2):
        x = x.float() if x.dtype != torch.float32 else x
        x = x if x.is_contiguous() else x.contiguous()
        B, S, D = x.shape
        mha = blk.self_attention.self_attention
        H = mha.num_heads
        dh = D // H
        Mh = blk.mlp.fc1.out_features
        R, dev, eps = B * S, x.device, blk.ln_1.eps
        bf = _lib.operand_dtype()
        h1 = torch.empty((R, D), dtype=bf, device=dev)
        qkv = torch.empty((R, 3 * D), dtype=bf, device=dev)
        att = torch.empty((R, D), dtype=bf, device=dev)
        x1 = torch.empty((B, S, D), dtype=torch.float32, device=dev)
        h2 = torch.empty((R, D), dtype=bf, device=dev)
        pair = torch.empty((R, 2 * Mh), dtype=bf, device=dev)           # [gelu(pre) | gelu'(pre)]: one fc1 epilogue writes both (round 6: the derivative, not the pre-activation)
        gl, pre = pair[:, :Mh], pair[:, Mh:]
        out = torch.empty_like(x)
        qscale = float(dh) ** -0.5
        ops.layernorm_bf16(x, _f32(ln1w), _f32(ln1b), eps, h1)
        ops.gemm(h1, bf16_weight(mha.in_proj_weight), _f32(inb), qkv, PV_EPI_BIAS_BF16, M=R, qcols=D, qscale=qscale)
        lse = _attn_lse(B, S, H, dh, dev)                    # row statistics for the persistent backward kernel (None: the two-pass kernel recomputes them)
        ops.attention(qkv, att, B, S, H, dh, lse=lse)
        ops.gemm(att, bf16_weight(mha.out_proj.weight), _f32(ob), x1.view(R, D), PV_EPI_BIAS_RES_F32, M=R, res=x.view(R, D))
        ops.layernorm_bf16(x1, _f32(ln2w), _f32(ln2b), blk.ln_2.eps, h2)
        ops.gemm(h2, bf16_weight(blk.mlp.fc1.weight), _f32(b1), pair, PV_EPI_BIAS_GELU_PAIR_BF16, M=R)
        ops.gemm(gl, bf16_weight(blk.mlp.fc2.weight), _f32(b2), out.view(R, D), PV_EPI_BIAS_RES_F32, M=R, res=x1.view(R, D))
        ctx.blk, ctx.dims = blk, (B, S, D, H, dh, Mh, qscale)
        ctx.tp, ctx.operand = current_pass(), _lib.current_operand()
        ctx.lse = lse
        ctx.save_for_backward(x, h1, qkv, att, x1, h2, pair)
        return out

    @staticmethod
    def backward(ctx, dout):
        with _kernels(ctx.tp, ctx.operand):
            return BlockFn._bw(ctx, dout)

    @staticmethod
    def _bw(ctx, dout):
        blk = ctx.blk
        x, h1, qkv, att, x1, h2, pair = ctx.saved_tensors
        B, S, D, H, dh, Mh, qscale = ctx.dims
        gl, pre = pair[:, :Mh], pair[:, Mh:]
        mha = blk.self_attention.self_attention
        R, dev, bf = B * S, x.device, _lib.operand_dtype()
        dout = dout.float() if dout.dtype != torch.float32 else dout
        ws = workspace

        # ---- MLP branch ------------------------------------------------------------------------------------
        dout3 = dout
        dout = (dout if dout.is_contiguous() else dout.contiguous()).view(R, D)
        # frozen parameters (the reference's finetuning trains gates / class tokens / head only, train/train.py:100): their weight-gradient
        # GEMMs - a fifth of the step - and bias sums are not computed.  needs_input_grad follows forward's argument order.
        need = dict(zip(("ln1w", "ln1b", "inw", "inb", "ow", "ob", "ln2w", "ln2b", "w1", "b1", "w2", "b2"), ctx.needs_input_grad[2:14]))
        d2, db2 = _bf16_grad(dout3, ws.get("bw_d", (R, D), bf, dev))
        db2_handed = db2 is not None
        dw2 = None
        if need["w2"]:
            dw2, db2c = _wgrad(d2, gl, "fc2", bias_grad=db2 is None)
            db2 = db2c if db2 is None else db2
        elif need["b2"] and db2 is None:
            db2 = ops.colsum(d2, torch.empty((D,), dtype=torch.float32, device=dev))
        dpre = ws.get("bw_dgl", (R, Mh), bf, dev)                               # (d2 . W2) * the saved gelu'(pre), fused in the epilogue
        db1 = torch.empty((Mh,), dtype=torch.float32, device=dev) if need["b1"] else None
        ops.gemm(d2, bf16_weight_t(blk.mlp.fc2.weight), None, dpre, PV_EPI_GELU_GRAD_BF16, M=R, res=pre, tag="[dgrad]", colsum_out=db1)
        dw1 = _wgrad(dpre, h2, "fc1", bias_grad=False)[0] if need["w1"] else None
        dhid = ws.get("bw_dh", (R, D), bf, dev)
        ops.gemm(dpre, bf16_weight_t(blk.mlp.fc1.weight), None, dhid, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        # (fp16 pass: the residual gradient at x1 travels to LayerNorm 1's backward as the 16-bit copy the GEMMs read anyway: _DX1_16 above)
        dx1_16 = _DX1_16 and bf == torch.float16
        dx1 = None if dx1_16 else ws.get("bw_dx1", (R, D), torch.float32, dev)
        dgb2 = torch.empty((3, D), dtype=torch.float32, device=dev)
        d1 = ws.get("bw_d1", (R, D), bf, dev)
        ops.layernorm_bwd(x1.view(R, D), dhid, _f32(blk.ln_2.weight), dout, dx1, dgb2, blk.ln_2.eps, dx_bf16=d1)
        # ---- attention branch ------------------------------------------------------------------------------
        dwo = _wgrad(d1, att, "proj", bias_grad=False)[0] if need["ow"] else None
        dbo = dgb2[2]                                                           # column sums of d1, from the LN2 backward pass
        datt = ws.get("bw_datt", (R, D), bf, dev)
        ops.gemm(d1, bf16_weight_t(mha.out_proj.weight), None, datt, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        dqkv = ws.get("bw_dqkv", (R, 3 * D), bf, dev)
        dbp = ws.get("bw_dbp", (B, 3 * D), torch.float32, dev) if need["inb"] else None
        if ctx.lse is not None:
            ops.attention_bwd_lse(qkv, datt, att, ctx.lse, dqkv, B, S, H, dh, qscale, dbias_partial=dbp)
        else:
            ops.attention_bwd(qkv, datt, dqkv, B, S, H, dh, qscale, dbias_partial=dbp)
        dbin = ops.colsum(dbp, torch.empty((3 * D,), dtype=torch.float32, device=dev)) if need["inb"] else None    # sum of the per-image column sums
        dwin = _wgrad(dqkv, h1, "qkv", bias_grad=False)[0] if need["inw"] else None
        ops.gemm(dqkv, bf16_weight_t(mha.in_proj_weight), None, dhid, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        dx = torch.empty((B, S, D), dtype=torch.float32, device=dev)
        dgb1 = torch.empty((3, D), dtype=torch.float32, device=dev)
        dxb = torch.empty((B, S, D), dtype=bf, device=dev)
        ops.layernorm_bwd(x.view(R, D), dhid, _f32(blk.ln_1.weight), d1 if dx1_16 else dx1, dx.view(R, D), dgb1, blk.ln_1.eps, dx_bf16=dxb)
        if debug_amax is not None:           # diagnostics (scripts/train_f16_probe.py): the largest 16-bit gradient of every kind in this block, in scaled units
            debug_amax.append({k: float(v.float().abs().max()) for k, v in (("d2", d2), ("dpre", dpre), ("d1", d1), ("datt", datt), ("dqkv", dqkv), ("dh1", dhid), ("dx", dxb))})
        if ctx.tp is not None:
            ctx.tp.note(dgb1)
        # parameter gradients leave the scaled chain (dbo = dgb2[2]; a handed-over db2 was unscaled by the block that produced it)
        _unscale(ctx.tp, dgb1, dgb2, dwin, dbin, dwo, dw1, db1, dw2, None if db2_handed else db2)
        dx._pv_bf16 = (dxb, dx._version, dgb1[2])    # hand-off to the previous block's backward (see _bf16_grad)
        return (None, dx, dgb1[0], dgb1[1], dwin, dbin, dwo, dbo, dgb2[0], dgb2[1], dw1, db1, dw2, db2)


class MaskedBlockFn(torch.autograd.Function):
    """ResidualViT's masked pre-LN block (reference models/residualvit.py:249-260): with a per-token mask m [B,S] (1 on the class /
    budget rows)  x1 = x + m * MHA(m * LN1(x)),  out = x1 + MLP(m * LN2(x1)).  Forward: the kernels of BlockFn with the row scale
    in the LayerNorm kernels and a masked residual add (the bf16 branch output u is kept for the mask gradient); backward:
    pv_layernorm_bwd_masked adds  dm = rowdot(dh1, LN1(x)) + rowdot(dh2, LN2(x1)) + rowdot(dx1, u)."""

    @staticmethod
    def forward(ctx, blk, x, m, ln1w, ln1b, inw, inb, ow, ob, ln2w, ln2b, w1, b1, w2, b2, h1_given=None):
        x = x.float() if x.dtype != torch.float32 else x
        x = x if x.is_contiguous() else x.contiguous()
        m = m.float().contiguous()
        B, S, D = x.shape
        mha = blk.self_attention.self_attention
        H = mha.num_heads
        dh = D // H
        Mh = blk.mlp.fc1.out_features
        R, dev, bf = B * S, x.device, _lib.operand_dtype()
        h1 = torch.empty((R, D), dtype=bf, device=dev)
        qkv = torch.empty((R, 3 * D), dtype=bf, device=dev)
        att = torch.empty((R, D), dtype=bf, device=dev)
        u = torch.empty((R, D), dtype=bf, device=dev)
        x1 = torch.empty((B, S, D), dtype=torch.float32, device=dev)
        h2 = torch.empty((R, D), dtype=bf, device=dev)
        pair = torch.empty((R, 2 * Mh), dtype=bf, device=dev)
        out = torch.empty_like(x)
        qscale = float(dh) ** -0.5
        mrow = m.view(R)
        if h1_given is not None and h1_given.shape == (R, D):
            h1 = h1_given                                    # m * LN1(x), emitted by the gate kernel
        else:
            ops.layernorm_bf16(x, _f32(ln1w), _f32(ln1b), blk.ln_1.eps, h1, mrow)
        ops.gemm(h1, bf16_weight(mha.in_proj_weight), _f32(inb), qkv, PV_EPI_BIAS_BF16, M=R, qcols=D, qscale=qscale)
        lse = _attn_lse(B, S, H, dh, dev)                    # row statistics for the persistent backward kernel (None: the two-pass kernel recomputes them)
        ops.attention(qkv, att, B, S, H, dh, lse=lse)
        ops.gemm(att, bf16_weight(mha.out_proj.weight), _f32(ob), u, PV_EPI_BIAS_BF16, M=R)
        ops.masked_residual(x.view(R, D), u, mrow, x1.view(R, D))
        ops.layernorm_bf16(x1, _f32(ln2w), _f32(ln2b), blk.ln_2.eps, h2, mrow)
        ops.gemm(h2, bf16_weight(blk.mlp.fc1.weight), _f32(b1), pair, PV_EPI_BIAS_GELU_PAIR_BF16, M=R)
        ops.gemm(pair[:, :Mh], bf16_weight(blk.mlp.fc2.weight), _f32(b2), out.view(R, D), PV_EPI_BIAS_RES_F32, M=R, res=x1.view(R, D))
        ctx.blk, ctx.dims = blk, (B, S, D, H, dh, Mh, qscale)
        ctx.tp, ctx.operand = current_pass(), _lib.current_operand()
        ctx.lse = lse
        ctx.save_for_backward(x, mrow, h1, qkv, att, u, x1, h2, pair)
        return out

    @staticmethod
    def backward(ctx, dout):
        with _kernels(ctx.tp, ctx.operand):
            return MaskedBlockFn._bw(ctx, dout)

    @staticmethod
    def _bw(ctx, dout):
        blk = ctx.blk
        x, mrow, h1, qkv, att, u, x1, h2, pair = ctx.saved_tensors
        B, S, D, H, dh, Mh, qscale = ctx.dims
        gl, pre = pair[:, :Mh], pair[:, Mh:]
        mha = blk.self_attention.self_attention
        R, dev, bf = B * S, x.device, _lib.operand_dtype()
        dout = dout.float() if dout.dtype != torch.float32 else dout
        ws = workspace
        dout3 = dout
        dout = (dout if dout.is_contiguous() else dout.contiguous()).view(R, D)
        # frozen block weights (the reference finetunes ResidualViT's gates / class tokens / head only, train/train.py:100): no weight-gradient GEMMs
        need = dict(zip(("ln1w", "ln1b", "inw", "inb", "ow", "ob", "ln2w", "ln2b", "w1", "b1", "w2", "b2"), ctx.needs_input_grad[3:15]))
        d2, db2 = _bf16_grad(dout3, ws.get("bw_d", (R, D), bf, dev))
        db2_handed = db2 is not None
        dw2 = None
        if need["w2"]:
            dw2, db2c = _wgrad(d2, gl, "fc2", bias_grad=db2 is None)
            db2 = db2c if db2 is None else db2
        elif need["b2"] and db2 is None:
            db2 = ops.colsum(d2, torch.empty((D,), dtype=torch.float32, device=dev))
        dpre = ws.get("bw_dgl", (R, Mh), bf, dev)
        db1 = torch.empty((Mh,), dtype=torch.float32, device=dev) if need["b1"] else None
        ops.gemm(d2, bf16_weight_t(blk.mlp.fc2.weight), None, dpre, PV_EPI_GELU_GRAD_BF16, M=R, res=pre, tag="[dgrad]", colsum_out=db1)
        dw1 = _wgrad(dpre, h2, "fc1", bias_grad=False)[0] if need["w1"] else None
        dhid = ws.get("bw_dh", (R, D), bf, dev)
        ops.gemm(dpre, bf16_weight_t(blk.mlp.fc1.weight), None, dhid, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        dx1 = ws.get("bw_dx1", (R, D), torch.float32, dev)
        du = ws.get("bw_d1", (R, D), bf, dev)
        dgb2 = torch.empty((3, D), dtype=torch.float32, device=dev)
        dm = torch.empty((R,), dtype=torch.float32, device=dev)
        ops.layernorm_bwd_masked(x1.view(R, D), dhid, _f32(blk.ln_2.weight), _f32(blk.ln_2.bias), mrow, dout, u, dx1, du, True, dgb2, dm, False,
                                 blk.ln_2.eps)
        dwo = _wgrad(du, att, "proj", bias_grad=False)[0] if need["ow"] else None
        dbo = dgb2[2]
        datt = ws.get("bw_datt", (R, D), bf, dev)
        ops.gemm(du, bf16_weight_t(mha.out_proj.weight), None, datt, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        dqkv = ws.get("bw_dqkv", (R, 3 * D), bf, dev)
        dbp = ws.get("bw_dbp", (B, 3 * D), torch.float32, dev) if need["inb"] else None
        if ctx.lse is not None:
            ops.attention_bwd_lse(qkv, datt, att, ctx.lse, dqkv, B, S, H, dh, qscale, dbias_partial=dbp)
        else:
            ops.attention_bwd(qkv, datt, dqkv, B, S, H, dh, qscale, dbias_partial=dbp)
        dbin = ops.colsum(dbp, torch.empty((3 * D,), dtype=torch.float32, device=dev)) if need["inb"] else None
        dwin = _wgrad(dqkv, h1, "qkv", bias_grad=False)[0] if need["inw"] else None
        ops.gemm(dqkv, bf16_weight_t(mha.in_proj_weight), None, dhid, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        dx = torch.empty((B, S, D), dtype=torch.float32, device=dev)
        dxb = torch.empty((B, S, D), dtype=bf, device=dev)
        dgb1 = torch.empty((3, D), dtype=torch.float32, device=dev)
        ops.layernorm_bwd_masked(x.view(R, D), dhid, _f32(blk.ln_1.weight), _f32(blk.ln_1.bias), mrow, dx1, None, dx.view(R, D), dxb, False, dgb1,
                                 dm, True, blk.ln_1.eps)
        if ctx.tp is not None:
            ctx.tp.note(dgb1)
        _unscale(ctx.tp, dgb1, dgb2, dwin, dbin, dwo, dw1, db1, dw2, None if db2_handed else db2)      # (dm stays inside the chain: it feeds GateFn)
        dx._pv_bf16 = (dxb, dx._version, dgb1[2])
        return (None, dx, dm.view(B, S), dgb1[0], dgb1[1], dwin, dbin, dwo, dbo, dgb2[0], dgb2[1], dw1, db1, dw2, db2, None)


class GateFn(torch.autograd.Function):
    """ResidualViT's sigmoid gate with the learnable budget threshold + the token masking it drives (models/residualvit.py:197-235,
    :47-74, models/blocks.py:62-69), forward and backward on one kernel each: tokens [B,S,D] -> (masked tokens [B,S,D], row_scale [B,S] =
    [1 | mask | 1], thresholds [B]).  row_scale carries the autograd graph of the mask: the masked block (MaskedBlockFn) and any auxiliary
    loss on `block.mask` (a view of it) send their gradients back through it."""

    @staticmethod
    def forward(ctx, x, wg, bg, wb, bb, temp, sbias, ln=None):
        """ln = (gamma, beta, eps) of the masked block's LN1 (detached): the kernel also emits h1 = row_scale * LN1(masked) from the registers
        the rows are in (a constant of this function as far as autograd goes: the block differentiates LN1 from its own input)."""
        x = x.float() if x.dtype != torch.float32 else x
        x = x if x.is_contiguous() else x.contiguous()
        masked = torch.empty_like(x)
        thr = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        h1 = torch.empty((x.shape[0] * x.shape[1], x.shape[2]), dtype=_lib.operand_dtype(), device=x.device) if ln is not None else None
        _mask, rs = ops.residual_gate(x, masked, _f32(wg).view(-1), _f32(bg), _f32(wb).view(-1), _f32(bb), temp, sbias, thr_out=thr,
                                      ln=None if ln is None else (_f32(ln[0]), _f32(ln[1]), float(ln[2]), h1))
        ctx.save_for_backward(x, wg, bg, wb, bb)
        ctx.cfg = (float(temp), float(sbias))
        ctx.tp, ctx.operand = current_pass(), _lib.current_operand()
        if h1 is None:
            h1 = torch.empty((0,), dtype=_lib.operand_dtype(), device=x.device)
        ctx.mark_non_differentiable(thr, h1)
        return masked, rs, thr, h1

    @staticmethod
    def backward(ctx, dmasked, drs, _dthr, _dh1):
        x, wg, bg, wb, bb = ctx.saved_tensors
        B, S, D = x.shape
        dmasked = torch.zeros_like(x) if dmasked is None else (dmasked.float() if dmasked.dtype != torch.float32 else dmasked).contiguous()
        drs = torch.zeros((B, S), dtype=torch.float32, device=x.device) if drs is None else drs.float().contiguous()
        with _kernels(ctx.tp, ctx.operand):
            dx, dwg, dbg, dwb, dbb = ops.residual_gate_bwd(x, dmasked, drs, _f32(wg).view(-1), _f32(bg), _f32(wb).view(-1), _f32(bb), *ctx.cfg)
        if ctx.tp is not None:
            ctx.tp.note(dwg)
        dbg, dbb = dbg.clone(), dbb.clone()        # (views of one 4-element reduction: unscaled as tensors of their own)
        _unscale(ctx.tp, dwg, dbg, dwb, dbb)       # dx stays in the chain
        return dx, dwg.view_as(wg), dbg.view_as(bg), dwb.view_as(wb), dbb.view_as(bb), None, None, None


def gate_forward_train(blk: nn.Module, x: torch.Tensor):
    """(masked tokens, row_scale [B,S], thresholds [B], h1 = row_scale * LN1(masked) as bf16 [B*S, D]) of a ResidualViT block with a sigmoid
    gate and the learnable budget token."""
    g, bgate = blk.residual_gate, blk.budget_token_gate
    return GateFn.apply(x, g.projection.weight, g.projection.bias, bgate.weight, bgate.bias, g.temp, g.sigmoid_bias,
                        (blk.ln_1.weight.detach(), blk.ln_1.bias.detach(), blk.ln_1.eps))


def masked_block_forward_train(blk: nn.Module, x: torch.Tensor, mask: torch.Tensor, h1: torch.Tensor = None) -> torch.Tensor:
    """mask: [B,S,1] or [B,S] (carries the gate's autograd graph); h1: mask * LN1(x) as bf16 [B*S, D] when the gate kernel produced it."""
    mha = blk.self_attention.self_attention
    m = mask.squeeze(-1) if mask.dim() == 3 else mask
    return MaskedBlockFn.apply(blk, x, m, blk.ln_1.weight, blk.ln_1.bias, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                               mha.out_proj.bias, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                               blk.mlp.fc2.weight, blk.mlp.fc2.bias, h1)


def block_forward_train(blk: nn.Module, x: torch.Tensor) -> torch.Tensor:
    mha = blk.self_attention.self_attention
    return BlockFn.apply(blk, x, blk.ln_1.weight, blk.ln_1.bias, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                         mha.out_proj.bias, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                         blk.mlp.fc2.weight, blk.mlp.fc2.bias)


class RowsBlockFn(torch.autograd.Function):
    """The LAST encoder block of a model forward in training, class-token row only: [B,S,D] -> [B,1,D] (engine.block_forward_rows is the
    inference counterpart).  The head reads that row alone (models/vit.py:242-246), so the gradient of every other output row is zero and
    the reference's backward through them multiplies zeros.  All-token work left: LN1, the k|v two thirds of the in-projection and their
    backward; everything else runs on B rows.
      forward   h1 = LN1(x) | kv = h1.Wkv^T+b | q = h1[cls].Wq^T+b (scaled) | att = attention_rows(q, kv) | x1 = x[cls] + att.Wo^T+b
                h2 = LN2(x1) | [gl | dgl] = fc1 pair | out = x1 + gl.W2^T+b                  saved: x, h1, kv, q, att, x1, h2, [gl | dgl]
      backward  as BlockFn on B rows down to datt; (dq, dkv) = attention_rows'(q, kv, att, datt); dWin = [dq^T.h1[cls] ; dkv^T.h1];
                dh1 = dkv.Wkv (+ dq.Wq on the class rows); dx = LN1'(dh1) (+ dx1 on the class rows)."""

    @staticmethod
    def forward(ctx, blk, x, m, ln1w, ln1b, inw, inb, ow, ob, ln2w, ln2b, w1, b1, w2, b2, h1_given=None):
        """m: None, or the ResidualViT row scale [B,S] of a masked block (h1 = m * LN1(x); the class-token rows carry scale 1, so everything
        behind the attention is the unmasked arithmetic)."""
        x = x.float() if x.dtype != torch.float32 else x
        x = x if x.is_contiguous() else x.contiguous()
        B, S, D = x.shape
        mha = blk.self_attention.self_attention
        H = mha.num_heads
        dh = D // H
        Mh = blk.mlp.fc1.out_features
        R, dev, eps, bf = B * S, x.device, blk.ln_1.eps, _lib.operand_dtype()
        qscale = float(dh) ** -0.5
        mrow = None if m is None else (m.float() if m.dtype != torch.float32 else m).contiguous().view(R)
        h1 = torch.empty((R, D), dtype=bf, device=dev)
        kv = torch.empty((R, 2 * D), dtype=bf, device=dev)
        q = torch.empty((B, D), dtype=bf, device=dev)
        att = torch.empty((B, D), dtype=bf, device=dev)
        x1 = torch.empty((B, D), dtype=torch.float32, device=dev)
        h2 = torch.empty((B, D), dtype=bf, device=dev)
        pair = torch.empty((B, 2 * Mh), dtype=bf, device=dev)
        gl = pair[:, :Mh]
        out = torch.empty((B, 1, D), dtype=torch.float32, device=dev)
        w_in = bf16_weight(mha.in_proj_weight)
        if h1_given is not None and h1_given.shape == (R, D):
            h1 = h1_given
        else:
            ops.layernorm_bf16(x, _f32(ln1w), _f32(ln1b), eps, h1, mrow)
        ops.gemm(h1, w_in[D:], _f32(inb)[D:], kv, PV_EPI_BIAS_BF16, M=R)
        ops.gemm(h1.view(B, S, D)[:, 0], w_in[:D], _f32(inb)[:D], q, PV_EPI_BIAS_BF16, M=B, qcols=D, qscale=qscale)
        ops.attention_rows(q, kv, att, B, S, 1, H, dh)
        ops.gemm(att, bf16_weight(mha.out_proj.weight), _f32(ob), x1, PV_EPI_BIAS_RES_F32, M=B, res=x[:, 0])
        ops.layernorm_bf16(x1, _f32(ln2w), _f32(ln2b), blk.ln_2.eps, h2)
        ops.gemm(h2, bf16_weight(blk.mlp.fc1.weight), _f32(b1), pair, PV_EPI_BIAS_GELU_PAIR_BF16, M=B)
        ops.gemm(gl, bf16_weight(blk.mlp.fc2.weight), _f32(b2), out.view(B, D), PV_EPI_BIAS_RES_F32, M=B, res=x1)
        ctx.blk, ctx.dims = blk, (B, S, D, H, dh, Mh, qscale)
        ctx.tp, ctx.operand = current_pass(), _lib.current_operand()
        ctx.masked = mrow is not None
        ctx.save_for_backward(x, h1, kv, q, att, x1, h2, pair, *([mrow] if mrow is not None else []))
        return out

    @staticmethod
    def backward(ctx, dout):
        with _kernels(ctx.tp, ctx.operand):
            return RowsBlockFn._bw(ctx, dout)

    @staticmethod
    def _bw(ctx, dout):
        blk = ctx.blk
        x, h1, kv, q, att, x1, h2, pair = ctx.saved_tensors[:8]
        mrow = ctx.saved_tensors[8] if ctx.masked else None
        B, S, D, H, dh, Mh, qscale = ctx.dims
        gl, pre = pair[:, :Mh], pair[:, Mh:]
        mha = blk.self_attention.self_attention
        R, dev, bf, f32 = B * S, x.device, _lib.operand_dtype(), torch.float32
        ws = workspace
        dout = (dout.float() if dout.dtype != f32 else dout).contiguous().view(B, D)
        # ---- MLP branch, class rows ---------------------------------------------------------------------------
        need = dict(zip(("ln1w", "ln1b", "inw", "inb", "ow", "ob", "ln2w", "ln2b", "w1", "b1", "w2", "b2"), ctx.needs_input_grad[3:15]))   # frozen: skipped
        d2 = ops.cast_bf16(dout, torch.empty((B, D), dtype=bf, device=dev))
        dw2, db2 = None, None
        if need["w2"]:
            dw2, db2 = _wgrad(d2, gl, "fc2")
        elif need["b2"]:
            db2 = dout.sum(0)
        dpre = torch.empty((B, Mh), dtype=bf, device=dev)
        db1 = torch.empty((Mh,), dtype=f32, device=dev) if need["b1"] else None
        ops.gemm(d2, bf16_weight_t(blk.mlp.fc2.weight), None, dpre, PV_EPI_GELU_GRAD_BF16, M=B, res=pre, tag="[dgrad]", colsum_out=db1)
        dw1 = _wgrad(dpre, h2, "fc1", bias_grad=False)[0] if need["w1"] else None
        dhq = torch.empty((B, D), dtype=bf, device=dev)
        ops.gemm(dpre, bf16_weight_t(blk.mlp.fc1.weight), None, dhq, PV_EPI_BIAS_BF16, M=B, tag="[dgrad]")
        dx1 = torch.empty((B, D), dtype=f32, device=dev)
        dgb2 = torch.empty((3, D), dtype=f32, device=dev)
        d1 = torch.empty((B, D), dtype=bf, device=dev)
        ops.layernorm_bwd(x1, dhq, _f32(blk.ln_2.weight), dout, dx1, dgb2, blk.ln_2.eps, dx_bf16=d1)
        # ---- attention branch ---------------------------------------------------------------------------------
        dwo = _wgrad(d1, att, "proj", bias_grad=False)[0] if need["ow"] else None
        datt = torch.empty((B, D), dtype=bf, device=dev)
        ops.gemm(d1, bf16_weight_t(mha.out_proj.weight), None, datt, PV_EPI_BIAS_BF16, M=B, tag="[dgrad]")
        dq = torch.empty((B, D), dtype=bf, device=dev)
        dkv = ws.get("bw_dqkv", (R, 2 * D), bf, dev)
        ops.attention_rows_bwd(q, kv, att, datt, dq, dkv, B, S, H, dh, qscale)
        dbin = None
        if need["inb"]:
            dbin = torch.empty((3 * D,), dtype=f32, device=dev)
            ops.colsum(dq, dbin[:D])
            ops.colsum(dkv, dbin[D:])
        h1c = h1.view(B, S, D)[:, 0]                                           # LN1 output of the class rows (row-strided view)
        dwin = None
        if need["inw"]:                                                        # frozen in the reference's gates-only finetuning
            dwin = torch.cat([_wgrad(dq, h1c, "q", bias_grad=False)[0], _wgrad(dkv, h1, "qkv", bias_grad=False)[0]], dim=0)
        wt = bf16_weight_t(mha.in_proj_weight)                                 # [D, 3D]
        dhid = ws.get("bw_dh", (R, D), bf, dev)
        ops.gemm(dkv, wt[:, D:], None, dhid, PV_EPI_BIAS_BF16, M=R, tag="[dgrad]")
        dhc = torch.empty((B, D), dtype=bf, device=dev)
        ops.gemm(dq, wt[:, :D], None, dhc, PV_EPI_BIAS_BF16, M=B, tag="[dgrad]")
        dhid.view(B, S, D)[:, 0] += dhc
        dx = torch.empty((B, S, D), dtype=f32, device=dev)
        dgb1 = torch.empty((3, D), dtype=f32, device=dev)
        dxb = torch.empty((B, S, D), dtype=bf, device=dev)
        dm = None
        if mrow is None:
            ops.layernorm_bwd(x.view(R, D), dhid, _f32(blk.ln_1.weight), None, dx.view(R, D), dgb1, blk.ln_1.eps, dx_bf16=dxb)
        else:                                                                  # h1 = m * LN1(x): the mask gradient falls out of the same pass
            dm = torch.empty((R,), dtype=f32, device=dev)
            ops.layernorm_bwd_masked(x.view(R, D), dhid, _f32(blk.ln_1.weight), _f32(blk.ln_1.bias), mrow, None, None, dx.view(R, D), dxb, False,
                                     dgb1, dm, False, blk.ln_1.eps)
            dm = dm.view(B, S)
        dx[:, 0] += dx1                                                        # the residual path exists for the class rows only
        dxb[:, 0] = dx[:, 0]
        dgb1[2] += dx1.sum(0)
        if ctx.tp is not None:
            ctx.tp.note(dgb1)
        _unscale(ctx.tp, dgb1, dgb2, dwin, dbin, dwo, dw1, db1, dw2, db2)
        dx._pv_bf16 = (dxb, dx._version, dgb1[2])
        return (None, dx, dm, dgb1[0], dgb1[1], dwin, dbin, dwo, dgb2[2], dgb2[0], dgb2[1], dw1, db1, dw2, db2, None)


def block_forward_rows_train(blk: nn.Module, x: torch.Tensor, mask: torch.Tensor = None, h1: torch.Tensor = None) -> torch.Tensor:
    mha = blk.self_attention.self_attention
    return RowsBlockFn.apply(blk, x, mask, blk.ln_1.weight, blk.ln_1.bias, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                             mha.out_proj.bias, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                             blk.mlp.fc2.weight, blk.mlp.fc2.bias, h1)


class EmbedFn(torch.autograd.Function):
    """Patch embedding + class/register tokens + positional embedding (reference models/vit.py:203-236, :92) with backward:
    dW_conv = dpatch^T . im2col(img), db_conv = colsum(dpatch), dpos = sum_b dtok[b], dcls = dpos[:n_cls]."""

    @staticmethod
    def forward(ctx, model, img, conv_w, conv_b, pos, cls, reg):
        from . import engine
        tokens = engine.embed_tokens(model, img)
        ctx.model = model
        ctx.tp, ctx.operand = current_pass(), _lib.current_operand()
        ctx.has_reg = reg is not None
        ctx.save_for_backward(img)
        return tokens

    @staticmethod
    def backward(ctx, dtok):
        with _kernels(ctx.tp, ctx.operand):
            return EmbedFn._bw(ctx, dtok)

    @staticmethod
    def _bw(ctx, dtok):
        model = ctx.model
        (img,) = ctx.saved_tensors
        dtok = dtok.float() if dtok.dtype != torch.float32 else dtok
        dtok = dtok if dtok.is_contiguous() else dtok.contiguous()
        B, S, D = dtok.shape
        P = model.patch_size
        ncls, nreg = model.num_class_tokens, model.num_registers
        nsp = ncls + nreg
        Np = S - nsp
        dev = dtok.device
        u8 = img.dtype == torch.uint8
        Cin = img.shape[3] if u8 else img.shape[1]
        K = Cin * P * P
        dwc, dbc = None, None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:          # a frozen stem (gates / class tokens / head finetuning) costs nothing here
            dpatch32 = dtok[:, nsp:, :].contiguous().view(B * Np, D)
            dpatch = ops.cast_bf16(dpatch32, workspace.get("bw_dgl", (B * Np, D), _lib.operand_dtype(), dev))
            if ctx.needs_input_grad[2]:
                cols = workspace.get("cols", (B * Np, K), _lib.operand_dtype(), dev)
                if u8:
                    from .engine import IMAGENET_MEAN, IMAGENET_STD
                    ops.im2col_u8(img, P, cols, getattr(model, "input_mean", IMAGENET_MEAN), getattr(model, "input_std", IMAGENET_STD))
                else:
                    ops.im2col(img if img.dtype == torch.float32 else img.float(), P, cols)
                dwc, dbc = _wgrad(dpatch, cols, "conv")
                dwc = dwc.view(model.conv_proj.weight.shape)
            else:
                dbc = ops.colsum(dpatch, torch.empty((D,), dtype=torch.float32, device=dev))
        dpos = ops.colsum(dtok.view(B, S * D), torch.empty((S * D,), dtype=torch.float32, device=dev)).view(1, S, D)
        if ctx.tp is not None:
            ctx.tp.note(dpos)                      # the chain's last word: anything non-finite upstream has reached the token gradient
        _unscale(ctx.tp, dwc, dbc, dpos)
        dcls = dpos[:, :ncls].clone()
        dreg = dpos[:, ncls:nsp].clone() if ctx.has_reg else None
        return (None, None, dwc, dbc, dpos, dcls, dreg)


def embed_tokens_train(model: nn.Module, img: torch.Tensor) -> torch.Tensor:
    if img.requires_grad:
        # someone differentiates with respect to the IMAGE (saliency maps, adversarial examples): EmbedFn has no col2im, so the stem runs on
        # the stock convolution (0.7 % of the FLOPs) and autograd delivers dL/d(image); the blocks behind it stay on the HIP functions
        return leave(model._composite_tokens(img.float() if img.dtype != torch.float32 else img) + model.encoder.pos_embedding, primary=True)
    reg = model.register_tokens if model.num_registers > 0 else None
    return EmbedFn.apply(model, img, model.conv_proj.weight, model.conv_proj.bias, model.encoder.pos_embedding, model.class_tokens, reg)


class SortDropFn(torch.autograd.Function):
    """RankViT ranking + compaction (models/rankvit.py:55-77) with the same HIP kernels as inference (bit-exact keep indices);
    backward scatters the gradient rows back to the kept positions (indices carry no gradient, as with torch.gather)."""

    @staticmethod
    def forward(ctx, x, budget):
        from . import engine
        x = x.float() if x.dtype != torch.float32 else x
        out, keep = engine.sort_and_drop(x, budget)
        ctx.s_in = x.shape[1]
        ctx.save_for_backward(keep)
        ctx.mark_non_differentiable(keep)
        return out, keep

    @staticmethod
    def backward(ctx, dout, _dkeep):
        (keep,) = ctx.saved_tensors
        dout = dout.float() if dout.dtype != torch.float32 else dout
        if keep.shape[1] == 0:                     # budget 0: only the class token went on
            dx = dout.new_zeros((dout.shape[0], ctx.s_in, dout.shape[2]))
            dx[:, :1] = dout
            return dx, None
        return ops.scatter_tokens(dout if dout.is_contiguous() else dout.contiguous(), keep, ctx.s_in), None


def sort_and_drop_train(x: torch.Tensor, budget: float):
    return SortDropFn.apply(x, budget)


def pool_and_head_train(model: nn.Module, tokens: torch.Tensor) -> torch.Tensor:
    """Final LayerNorm on the class-token rows, sum, head: [B, n_cls, D] fp32 stock ops under autograd (0.02 % of the
    step's FLOPs; the gradient re-enters the HIP backward as dL/d(tokens), zero outside the class rows)."""
    tokens = enter(tokens, primary=True)              # stock autograd ends here: the gradient enters the (scaled) HIP chain
    cls = model.encoder.ln(tokens[:, 0:model.num_class_tokens])
    return model.head(cls.sum(dim=1))


def train_eligible(x: torch.Tensor, module: nn.Module, dropout_p: float) -> bool:
    from . import engine
    return (x.is_cuda and x.numel() > 0 and torch.is_grad_enabled() and not (module.training and dropout_p > 0.0) and engine._PRECISION in ("auto", "bf16")
            and os.environ.get("PEEKVIT_AMD_BACKEND", "") != "torch" and os.environ.get("PEEKVIT_AMD_TRAIN", "hip") == "hip")
