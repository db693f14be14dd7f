"""Mode "auto" chooses hipGraph replay BY ITSELF for launch-bound model forwards (round 6; `graph.GraphedForward` stays the explicit form).

vit_tiny at batch 32 (BASELINE config 1's model) is ~100 launches of a few microseconds each: 0.59 ms eager, 0.41 ms as one hipGraph replay, bit-identical
(round 5 bench entry `vit_tiny_fwd_b32`).  Nothing picked the replay unless the user wrapped the model.  Now `engine.run_guarded` - the one door every
model-level inference forward goes through - asks this module first:

* a forward is a CANDIDATE when its estimated MFMA work is small against the host cost of its launches (`launch_bound`), it runs on the main thread in
  precision mode "auto", outside `engine.deferred_flags()` and stream capture, nobody watches it through module hooks, and its input lives on a GPU;
* the first `WARM` candidate forwards of a key run eagerly as before (the self-check verdict of the key must be "ok" and no guard may trip); then ONE capture
  (`graph.GraphedForward`: private workspace arena, side stream) and replays from there on: input copied into the graph's static tensor, replay, the guard word
  read (the same one host synchronisation the eager forward makes), the logits CLONED out of the static output;
* a replay whose guard word is not clean is thrown away and the forward runs eagerly on the caller's tensor (repeats, local fallback and sticky verdicts are
  `run_guarded`'s business); replays advance the periodic self-check's counter and the forward on which a probe is due runs eagerly;
* a graph is valid for what it was captured under and is dropped the moment any of it differs: the module's guard state (kept IN the `GuardState`, which an
  optimizer step or `load_state_dict` resets), every parameter's version counter, the identity of the encoder's layers, a module / parameter registered
  anywhere since (global registration hooks), module hooks, the budget setting (part of the key) and the engine's knobs.

`PEEKVIT_AMD_AUTO_GRAPH=0` switches it off.  Anything unexpected during capture switches it off for that key, never the forward."""
from __future__ import annotations

import os
import threading
from typing import Optional

import torch
import torch.nn as nn

ENABLED = os.environ.get("PEEKVIT_AMD_AUTO_GRAPH", "1") != "0"
WARM = int(os.environ.get("PEEKVIT_AMD_AUTO_GRAPH_WARM", "4"))            # clean eager forwards of a key before its capture
MAX_GRAPHS = 4                                                              # per module (least recently captured goes first)
# host time of one eager forward: ~0.12 ms of model-level Python + ~0.115 ms per encoder layer (vit_tiny, 4 layers: 0.586 ms per forward at batch 32 with the
# GPU idle a third of the time, round 5 bench) against ~300 TFLOP/s of small-shape GEMM throughput: a forward whose GEMM FLOPs take less than that gains
HOST_MS_BASE, HOST_MS_PER_LAYER = 0.12, 0.115
SMALL_TFLOPS = 300.0

replays = 0            # counters (tests / bench)
captures = 0
drops = 0

_structure_epoch = 0   # any module / parameter / buffer registered anywhere in the process advances it


def _bump(*_a, **_k):
    global _structure_epoch
    _structure_epoch += 1
    return None


try:
    import torch.nn.modules.module as _m
    _m.register_module_module_registration_hook(_bump)
    _m.register_module_parameter_registration_hook(_bump)
    _m.register_module_buffer_registration_hook(_bump)
except (ImportError, AttributeError):                 # pragma: no cover - a torch without the registration hooks: structure changes are not seen -> no auto graphs
    ENABLED = False


def _knobs():
    from . import engine, ops
    return (engine._STREAMS, engine.LOCAL_FALLBACK, engine.SELFCHECK_IMAGES, engine.SELFCHECK_LIMIT, engine.RANK_STRICT, engine._FUSE_LN, engine._FULLROW_LN,
            engine._FOLD_LN, engine._FUSE_RANK_NORM, engine._SMALL_M_SPLITK, engine._LAST_BLOCK_ROWS, engine._GATE_NO_MASKED, engine.FALLBACK_MODE,
            engine.RANK_REPAIR, engine.RANK_TIE_GAP, engine._FUSED_PATCH_EMBED,
            getattr(ops, "knob_epoch", 0))


def launch_bound(owner: nn.Module, batch: int) -> bool:
    """Is a forward of `owner` at this batch size bound by the host's launches rather than by the GPU?  GEMM + attention FLOPs of the encoder from the
    module's own attributes (the formula of SURVEY.md section 8d) against the host time of an eager forward."""
    try:
        D, L, S = int(owner.hidden_dim), len(owner.encoder.layers), int(owner.seq_length)
        Mh = int(getattr(owner, "mlp_dim", 4 * D))
    except (AttributeError, TypeError):
        return False
    flops = batch * L * (S * D * 3 * D * 2 + 2 * S * S * D * 2 + S * D * D * 2 + 2 * S * D * Mh * 2)
    return flops / (SMALL_TFLOPS * 1e12) < (HOST_MS_BASE + HOST_MS_PER_LAYER * L) * 1e-3


class _Entry:
    __slots__ = ("warm", "graph", "sig", "params", "mods", "calls", "dead")

    def __init__(self):
        self.warm, self.graph, self.sig, self.params, self.mods, self.calls, self.dead = 0, None, None, None, None, 0, False


_version_of = __import__("operator").attrgetter("_version")


def _signature(owner: nn.Module, ent: _Entry, st) -> tuple:
    from . import engine
    layers = getattr(getattr(owner, "encoder", None), "layers", ())
    return (_structure_epoch, engine._opt_generation, tuple(map(id, layers)), sum(map(_version_of, ent.params)), _knobs(),
            st.unsafe, st.no_fold, st.hybrid, st.mlp_hybrid, owner.training)


def _hooked(ent: _Entry) -> bool:
    import torch.nn.modules.module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    # (the modules' hook dictionaries themselves, collected once per entry: a registration mutates them in place; `any` over ~100 dicts runs in C -
    #  this check is on the replay path of a 0.4 ms forward)
    if ent.mods and not isinstance(ent.mods[0], dict):
        ent.mods = [d for mod in ent.mods for d in (mod._forward_hooks, mod._forward_pre_hooks)]
    return any(ent.mods)


def _eligible(owner: nn.Module, x: torch.Tensor) -> bool:
    from . import engine, ops
    # (a per-launch KernelTimer wants every launch on the stream, with its two events around it: neither capture nor replay while one is active)
    return (ENABLED and ops._timer is None and x.is_cuda and not torch.is_grad_enabled() and threading.current_thread() is threading.main_thread()
            and getattr(engine._region, "defer", None) is None and not getattr(engine._region, "autograph_busy", False)
            and not getattr(engine._region, "in_probe", False) and not torch.cuda.is_current_stream_capturing())


def _entry(owner: nn.Module, x: torch.Tensor, probe_key, st, create: bool) -> Optional[_Entry]:
    graphs = st.graphs
    key = (probe_key, tuple(x.shape), x.dtype, x.device.index)
    ent = graphs.get(key)
    if ent is None and create:
        if getattr(owner, "_pv_no_autograph", False) or not launch_bound(owner, int(x.shape[0])):      # (a forward with a host decision inside: RankViT's near-tie repair)
            return None
        while len(graphs) >= MAX_GRAPHS:
            graphs.pop(next(iter(graphs)))
        ent = graphs[key] = _Entry()
    return ent


def try_replay(owner: nn.Module, x: torch.Tensor, probe_key, st) -> Optional[torch.Tensor]:
    """The logits of `x` from the key's captured graph, or None (no graph, no longer valid, a guard bit raised, or the periodic eager forward is due)."""
    global replays, drops
    if not st.graphs or not _eligible(owner, x):
        return None
    ent = _entry(owner, x, probe_key, st, create=False)
    if ent is None or ent.graph is None:
        return None
    from . import engine
    if _hooked(ent) or _signature(owner, ent, st) != ent.sig:
        st.graphs.pop((probe_key, tuple(x.shape), x.dtype, x.device.index), None)
        drops += 1
        return None
    ent.calls += 1
    if engine.SELFCHECK_EVERY > 0:
        # the periodic self-check lives in the eager path and counts EVERY guarded forward of the key, replayed or not: a replay advances the same
        # counter, and the forward on which the probe is due is left to the eager path (which counts it and probes)
        vkey = (probe_key, int(x.shape[0]), st.no_fold, x.dtype, tuple(x.shape[1:]), st.hybrid, st.mlp_hybrid)
        n = st.calls.get(vkey, 0) + 1
        if n >= engine.SELFCHECK_EVERY:
            return None
        st.calls[vkey] = n
    g = ent.graph
    g.static_in.copy_(x)
    g.graph.replay()
    if engine._flag_bits(g._flag.tolist()) != 0:               # (the forward's one host synchronisation, as in the eager path)
        st.graphs.pop((probe_key, tuple(x.shape), x.dtype, x.device.index), None)
        drops += 1
        return None                                             # the eager path repeats this batch and draws the consequences
    replays += 1
    engine._region.last = "guarded"
    return g.static_out.clone()


def note_clean_eager(owner: nn.Module, x: torch.Tensor, probe_key, st, verdict_ok: bool):
    """A guarded eager forward of this key returned with a clean guard word: count it, and capture once the key is warm."""
    global captures
    if not _eligible(owner, x) or not verdict_ok:
        return
    ent = _entry(owner, x, probe_key, st, create=True)
    if ent is None or ent.dead or ent.graph is not None:
        return
    if ent.params is None:
        ent.params = list(owner.parameters())
        ent.mods = list(owner.modules())
    sig = _signature(owner, ent, st)
    if sig != ent.sig:
        ent.sig, ent.warm = sig, 0
    if _hooked(ent):
        ent.warm = 0
        return
    ent.warm += 1
    if ent.warm < WARM:
        return
    from . import engine
    from .graph import GraphedForward
    # (this runs at the tail of the guarded region of the forward that completed the warm-up: the capture's own forwards must enter `run_guarded` as
    #  top-level calls - inside an active region a forward runs unguarded on whatever operand library is current, i.e. bf16)
    engine._region.autograph_busy = True
    was_active, engine._region.active = getattr(engine._region, "active", False), False
    try:
        with torch.inference_mode(False):          # (the graph's static tensors outlive this call: never inference tensors, which refuse in-place updates outside inference mode)
            g = GraphedForward(owner, x, warmup=1, capture_error_mode="thread_local")
        if not g._guarded:
            ent.dead = True                                     # (the capture did not end in the guarded fp16 forward: nothing to replay safely)
            return
        ent.params, ent.mods = list(owner.parameters()), list(owner.modules())
        ent.graph, ent.sig, ent.calls = g, _signature(owner, ent, st), 0
        captures += 1
    except Exception as e:                                      # noqa: BLE001 - a failed capture must never fail the forward that triggered it
        ent.dead = True
        engine._warn_once(f"autograph:{id(owner)}", f"peekvit_amd: hipGraph capture of this launch-bound forward failed ({type(e).__name__}: {e}); it stays eager")
    finally:
        engine._region.autograph_busy = False
        engine._region.active = was_active
