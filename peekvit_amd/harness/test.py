"""Evaluation entry point, counterpart of validate/test.py:35-179: per budget AND per noise value (test.py:72-111: a NoiseBlock
spliced into the encoder, its SNR / drop probability swept) - accuracy, images/sec with the REFERENCE's definition (len(dataset) /
wall time of the whole loader loop incl. host->device copies, test.py:113-124) next to pure device time, FLOPs with the reference's
conventions (peekvit_amd.flops) and sparsity.

    python -m peekvit_amd.harness.test model=vit_b_16 test.test_batch_size=2048 dataset.val_size=4096 device=cuda:0
    python -m peekvit_amd.harness.test model=vit_tiny noise=gaussian test.noises=[0.0,5,20] device=cpu
"""
from __future__ import annotations

import json
import sys
import time
from typing import List, Sequence

import torch
from torch.utils.data import DataLoader

from .. import flops
from . import checkpoint
from .config import instantiate, load_config


@torch.no_grad()
def _resolve_timed(out, events):
    """engine.resolve(out); a batch that tripped a guard is repeated in there - that forward belongs to the loop's device time too (round-4 review:
    images/sec was overstated when a guard tripped)."""
    from .. import engine
    n0 = engine.fallback_count + engine.fold_fallback_count + engine.hybrid_fallback_count
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    res = engine.resolve(out)
    if engine.fallback_count + engine.fold_fallback_count + engine.hybrid_fallback_count != n0:
        e1.record()
        events.append((e0, e1))
    return res


def evaluate(model, loader, device, budgets: Sequence, n_images: int, prefetch: bool = True, noise_module=None,
             noise_vals: Sequence = (None,)) -> List[dict]:
    """prefetch (GPU only): batches are copied to the device one ahead of the forward on a side stream (harness.pipeline) and nothing is
    read back per batch - the loop body is the reference's, the host just never waits inside it.  prefetch=False is the reference's
    loop verbatim (synchronous copy, one .item() per batch).
    noise_module / noise_vals: the reference's inner loop (test.py:98-104) - for every budget, every value is set on the spliced
    NoiseBlock before the loader is swept; FLOPs are taken once per budget (the noise does not change them)."""
    model.eval().to(device)
    device = torch.device(device)
    results = []
    for budget in budgets:
        if budget is not None and hasattr(model, "set_budget"):
            model.set_budget(budget)
        fl = sparsity = None
        for noise_val in noise_vals:
            if noise_module is not None:
                noise_module.set_value(noise_val)
            correct, dev_ms, events = 0, 0.0, []
            start = time.time()
            n_batches = 0
            if device.type == "cuda" and prefetch:
                from .pipeline import DevicePrefetcher
                from .. import engine
                hits = torch.zeros((), dtype=torch.int64, device=device)
                prev = None
                # the guard word of batch i is read after batch i + 1 has been launched (engine.deferred_flags): no host stall per batch
                with engine.deferred_flags():
                    for batch, labels in DevicePrefetcher(loader, device, keep=1):       # (batch i stays valid through iteration i + 1: resolve() may repeat it)
                        n_batches += 1
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        out = model(batch)
                        e1.record()
                        events.append((e0, e1))
                        if prev is not None:
                            hits += (_resolve_timed(prev[0], events).argmax(1) == prev[1]).sum()
                        prev = (out, labels)
                    if prev is not None:
                        hits += (_resolve_timed(prev[0], events).argmax(1) == prev[1]).sum()
                correct = int(hits.item())                         # the one read-back of the loop
                dev_ms = sum(a.elapsed_time(b) for a, b in events)
            else:
                for batch, labels in loader:
                    n_batches += 1
                    batch, labels = batch.to(device), labels.to(device)
                    if device.type == "cuda":
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    out = model(batch)
                    if device.type == "cuda":
                        e1.record()
                        e1.synchronize()
                        dev_ms += e0.elapsed_time(e1)
                    correct += int((out.argmax(1) == labels).sum().item())
            elapsed = time.time() - start
            if n_batches == 0:
                raise ValueError("evaluate(): the loader yielded no batch")
            if fl is None:
                if noise_module is not None:                       # the FLOP pass sees the clean channel (snr 0 / prob 0 add nothing)
                    noise_module.set_value(0)
                fl, sparsity = flops.measured_flops(model, batch)
                if noise_module is not None:
                    noise_module.set_value(noise_val)
            row = {"budget": budget, "accuracy": correct / n_images, "images_per_second": n_images / elapsed,
                   "device_images_per_second": n_images / (dev_ms * 1e-3) if dev_ms else None,
                   "flops_per_image": fl, "sparsity": sparsity}
            if noise_module is not None:
                row["noise_type"], row["noise"] = noise_module.noise_type, noise_val
            results.append(row)
    return results


def main(argv: Sequence[str] = ()) -> List[dict]:
    cfg = load_config("test_config", list(argv))
    torch.manual_seed(cfg["seed"])
    device = torch.device(cfg["device"])
    dataset = instantiate(cfg["dataset"])
    loader = DataLoader(dataset.val_dataset, batch_size=cfg["test"]["test_batch_size"], shuffle=False,
                        num_workers=cfg["test"].get("num_workers", 0), pin_memory=device.type == "cuda")
    if cfg.get("load_from"):
        model, _ = checkpoint.load_state(checkpoint.get_checkpoint_path(cfg["load_from"]))
    else:
        model = instantiate(cfg["model"])
    budgets = cfg["test"].get("budgets") or [None]
    noise_module, noise_vals = None, [None]
    if cfg.get("noise"):                                       # test.py:72-78: `noise=gaussian|digital` splices the block, test.noises is swept
        from .noise import add_noise
        ns = dict(cfg["noise"])
        noise_module = add_noise(model, layer=ns.pop("layer"), noise_type=ns.pop("noise_type"), **ns)
        noise_vals = cfg["test"].get("noises") or [0.0]
    results = evaluate(model, loader, device, budgets, len(dataset.val_dataset), noise_module=noise_module, noise_vals=noise_vals)
    for r in results:
        print(json.dumps(r))
    return results


if __name__ == "__main__":
    main(sys.argv[1:])
