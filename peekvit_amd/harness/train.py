"""Training entry point, counterpart of train/train.py:34-211: epochs of forward -> CrossEntropy -> backward ->
[data-parallel gradient all-reduce] -> clip_grad_norm_(1.0) -> Adam step, validation and reference-format checkpoints.
On a GPU the step runs on the MI355X kernels end to end: forward with saved activations and the hand-written backward
(peekvit_amd.train_engine, autograd Functions), parameter gradients in fp32; PEEKVIT_AMD_TRAIN=torch selects the stock-op composite.

    python -m peekvit_amd.harness.train model=vit_tiny training.num_epochs=1 device=cpu
    torchrun --nproc-per-node 8 -m peekvit_amd.harness.train model=vit_b_16 training.train_batch_size=1024 device=cuda
"""
from __future__ import annotations

import json
import os
import sys
from typing import Sequence

import torch
import torch.distributed as td
from torch.utils.data import DataLoader

from .. import dist as pdist
from . import checkpoint
from .config import instantiate, load_config
from .test import evaluate


def train_epoch(model, loader, optimizer, device, clip: float, distributed: bool) -> float:
    model.train()
    loss_fn = torch.nn.CrossEntropyLoss()
    last = float("nan")
    reducer = pdist.OverlappedGradReducer(model.parameters(), model=model) if distributed else None   # buckets leave while backward still runs
    device = torch.device(device)
    if device.type == "cuda":                              # host -> HBM copy of the next batch under this step (harness.pipeline)
        from .pipeline import DevicePrefetcher
        loader = DevicePrefetcher(loader, device)
    for batch, labels in loader:
        batch, labels = batch.to(device), labels.to(device)
        if distributed:                                   # rank r trains on samples r::world of the global batch
            batch, labels = pdist.shard_batch(batch), pdist.shard_batch(labels)
        optimizer.zero_grad()
        loss = loss_fn(model(batch), labels)
        loss.backward()
        if distributed:
            reducer.finish()                               # every bucket reduced before the global-norm clip (train.py:120-121)
        if distributed and reducer.skip_step:
            continue                                       # an fp16 gradient overflowed on some rank (train_engine: loss scaling): every rank skips alike
        if clip:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        optimizer.step()                                   # (single process: a skipped step has every .grad = None - the optimizer passes over them)
        last = loss.detach()
    last = float(last) if torch.is_tensor(last) else last   # one read-back per epoch, not one per step
    if reducer is not None:
        reducer.remove()
    return last


def main(argv: Sequence[str] = ()) -> dict:
    cfg = load_config("train_config", list(argv))
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    device = torch.device(cfg["device"])
    if distributed:
        td.init_process_group("nccl" if device.type == "cuda" else "gloo")
        if device.type == "cuda":
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
            torch.cuda.set_device(device)
    torch.manual_seed(cfg["seed"])
    dataset = instantiate(cfg["dataset"])
    tr = cfg["training"]
    loader = DataLoader(dataset.train_dataset, batch_size=tr["train_batch_size"], shuffle=False, num_workers=tr.get("num_workers", 0))
    val_loader = DataLoader(dataset.val_dataset, batch_size=tr["eval_batch_size"], shuffle=False)
    model = instantiate(cfg["model"]).to(device)
    torch.nn.init.normal_(model.head.weight, std=0.02)        # the reference's zero head gives zero gradients to the trunk at step 0
    optimizer = instantiate(cfg["optimizer"], params=model.parameters())
    history = {"loss": [], "val_accuracy": []}
    model_args = dict(cfg["model"])                            # as the reference stores it (train.py: dict(cfg.model), `_target_` included)
    start_epoch = 0
    if cfg.get("load_from"):
        # the reference's semantics (train/train.py:64-70): INITIALISE THE WEIGHTS from a finished run - a .pth file or an experiment
        # directory (its last checkpoint) - with strict=False; optimizer state and epoch are ignored and training runs epochs 0..N.
        # This is how a ResidualViT / RankViT is fine-tuned from a trained ViT (different parameter sets, so the stored optimizer
        # state could not be loaded anyway).
        lf = cfg["load_from"]
        ck = lf if str(lf).endswith(".pth") else checkpoint.get_checkpoint_path(lf)
        if ck is None or not os.path.isfile(ck):
            raise FileNotFoundError(f"load_from={lf!r}: no checkpoint found (a .pth file, or a directory with checkpoints/*.pth)")
        print("Loading model from checkpoint: ", ck)
        checkpoint.load_state(ck, model=model)
    if cfg.get("resume_from"):
        # NOT in the reference: continue an interrupted run of THIS model - weights, optimizer state and epoch of the last checkpoint
        rf = cfg["resume_from"]
        ck = rf if str(rf).endswith(".pth") else checkpoint.get_checkpoint_path(rf)
        if ck is None or not os.path.isfile(ck):
            raise FileNotFoundError(f"resume_from={rf!r}: no checkpoint found")
        _, st = checkpoint.load_state(ck, model=model, optimizer=optimizer, strict=True)
        start_epoch = int(st.get("epoch", -1)) + 1
        if start_epoch >= tr["num_epochs"]:
            print(f"resume_from: checkpoint {ck} is of epoch {start_epoch - 1}, training.num_epochs={tr['num_epochs']}: nothing left to train")
    for epoch in range(start_epoch, tr["num_epochs"]):
        history["loss"].append(train_epoch(model, loader, optimizer, device, tr.get("clip_grad_norm", 1.0), distributed))
        if (epoch + 1) % tr.get("eval_every", 1) == 0:
            budget = [1.0] if hasattr(model, "set_budget") and not getattr(model, "add_budget_token", False) else [None]
            history["val_accuracy"].append(evaluate(model, val_loader, device, budget, len(dataset.val_dataset))[0]["accuracy"])
        if cfg.get("experiment_dir") and (not distributed or td.get_rank() == 0):
            history["checkpoint"] = checkpoint.save_state(cfg["experiment_dir"], model, model_args, optimizer=optimizer, epoch=epoch)
    if not distributed or td.get_rank() == 0:
        print(json.dumps(history))
    if distributed:
        td.destroy_process_group()
    return history


if __name__ == "__main__":
    main(sys.argv[1:])
