"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Run in the build container only (the reference never travels to the GPU box):

    python oracle/make_golden.py

Recipe: the reference must be importable as `peekvit`, so a symlink
/tmp/oracle_ref/peekvit -> /root/reference is put on sys.path; `torchvision` (absent here, used by
the reference only inside its pretrained-weight download branch, models/vit.py:278) is replaced by
three empty placeholder modules; bytecode writing is disabled because the reference is read-only.
Inputs and weights come from peekvit_amd.synth (pure functions of name+seed, bf16-representable),
so fixtures hold only OUTPUTS (+ a few small inputs) and stay tiny.
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from peekvit_amd import synth

GOLD = os.path.join(REPO, "tests", "golden")


REF_ROOT = "/root/reference"
REF_FILES = ("models/blocks.py", "models/vit.py", "models/rankvit.py", "models/residualvit.py", "models/adapters.py")


def _is_repo_path(p):
    try:
        return os.path.realpath(p or os.getcwd()) == os.path.realpath(REPO)
    except OSError:
        return False


def reference_sha256():
    import hashlib
    return {f: hashlib.sha256(open(os.path.join(REF_ROOT, f), "rb").read()).hexdigest() for f in REF_FILES}


def import_reference():
    """Import the REAL reference's three model classes; refuses to return anything that does not live under /root/reference.

    The build ships its own regular package `peekvit/` (the `_target_` alias boundary) at the repo root; a regular package beats the
    namespace portion /tmp/oracle_ref/peekvit -> /root/reference on sys.path, so the repo root (and '' when the cwd is the repo) is
    taken OFF sys.path and every already-imported `peekvit` / `peekvit.*` module is purged before the reference is imported.
    `peekvit_amd` (synth) is imported above, before this runs, and stays importable from sys.modules.
    """
    import inspect
    if not os.path.isdir(REF_ROOT):
        raise SystemExit("reference checkout not present: golden vectors can only be made in the build container")
    link_dir = "/tmp/oracle_ref"
    os.makedirs(link_dir, exist_ok=True)
    link = os.path.join(link_dir, "peekvit")
    if not os.path.islink(link):
        os.symlink(REF_ROOT, link)
    for name in [n for n in sys.modules if n == "peekvit" or n.startswith("peekvit.")]:
        del sys.modules[name]
    sys.path[:] = [p for p in sys.path if not _is_repo_path(p)]
    sys.path.insert(0, link_dir)
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvv = types.ModuleType("torchvision.models.vision_transformer")
    tvv.ViT_B_16_Weights = None
    tvv.ViT_B_32_Weights = None
    tv.models, tvm.vision_transformer = tvm, tvv
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.vision_transformer": tvv})
    from peekvit.models.vit import VisionTransformer
    from peekvit.models.rankvit import RankVisionTransformer
    from peekvit.models.residualvit import ResidualVisionTransformer
    classes = (VisionTransformer, RankVisionTransformer, ResidualVisionTransformer)
    for cls in classes:
        src = os.path.realpath(inspect.getsourcefile(cls))
        if not src.startswith(REF_ROOT + "/"):
            raise SystemExit(f"import_reference resolved {cls.__name__} to {src}, not the reference: refusing to write fixtures")
    return classes


def _cls_hooks(model, store):
    hs = []
    for blk in model.encoder.layers:
        hs.append(blk.register_forward_hook(lambda m, i, o: store.append(o[:, 0].detach().clone())))
    return hs


def run_vit(VT, name, batch, train_mode=False, full=False):
    cfg = synth.MODEL_CONFIGS[name]
    torch.manual_seed(0)
    m = VT(**cfg)
    synth.load_synth_weights(m, cfg, "vit", seed=0)
    m.train(train_mode)
    x = torch.from_numpy(synth.synth_images(batch, cfg["image_size"], seed=0))
    cls_rows, outs = [], []
    hooks = _cls_hooks(m, cls_rows)
    if full:
        for blk in m.encoder.layers:
            hooks.append(blk.register_forward_hook(lambda mod, i, o: outs.append(o.detach().clone())))
    enc = []
    hooks.append(m.encoder.register_forward_hook(lambda mod, i, o: enc.append(o.detach().clone())))
    tok_in = []
    hooks.append(m.encoder.register_forward_pre_hook(lambda mod, i: tok_in.append(i[0].detach().clone())))
    t0 = time.time()
    with torch.no_grad():
        logits = m(x)
    dt = time.time() - t0
    for h in hooks:
        h.remove()
    d = dict(logits=logits.numpy(), encoder_cls=enc[0][:, 0].numpy(),
             block_cls=torch.stack(cls_rows).numpy())
    if full:
        d["tokens_prepos"] = tok_in[0].numpy()      # encoder input, before + pos_embedding
        d["block_out"] = torch.stack(outs).numpy()
        d["encoder_out"] = enc[0].numpy()
    print(f"  {name} B={batch} train={train_mode}: {dt*1e3:.0f} ms, |logits| {logits.abs().mean():.4f}")
    return d


def sorted_gap_tokens(B, N, D, seed):
    """Tokens whose L2 norms are a (hashed) permutation of 1.0 + 0.01*i: gaps >> fp32 rounding."""
    dirs = synth.tensor(f"rank/dirs/{N}x{D}", (B, N, D), "normal", 1.0, 0.0, seed, bf16=False).astype(np.float64)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    order = np.argsort(synth.hash_uniform(f"rank/perm/{N}", B * N, seed).reshape(B, N), axis=-1)
    norms = 1.0 + 0.01 * order
    cls = synth.tensor(f"rank/cls/{D}", (B, 1, D), "normal", 1.0, 0.0, seed)
    tok = synth.round_to_bf16((dirs * norms[..., None]).astype(np.float32))
    return np.concatenate([cls, tok], axis=1)


def main():
    os.makedirs(GOLD, exist_ok=True)
    VT, RVT, ResVT = import_reference()
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__, "reference": "alessiodevoto/peekvit @ 2024_08_07", "reference_sha256": reference_sha256(),
            "errors": {}}

    # (1)-(3) plain ViT: micro (full tensors), tiny / small / B16 (logits + CLS rows), train-mode == eval-mode
    print("plain ViT")
    np.savez_compressed(os.path.join(GOLD, "vit_micro.npz"), **run_vit(VT, "vit_micro", 2, full=True))
    np.savez_compressed(os.path.join(GOLD, "vit_micro_train.npz"), **run_vit(VT, "vit_micro", 2, train_mode=True))
    np.savez_compressed(os.path.join(GOLD, "vit_tiny.npz"), **run_vit(VT, "vit_tiny", 2))
    np.savez_compressed(os.path.join(GOLD, "vit_tiny_train.npz"), **run_vit(VT, "vit_tiny", 2, train_mode=True))
    np.savez_compressed(os.path.join(GOLD, "vit_small.npz"), **run_vit(VT, "vit_small", 2))
    np.savez_compressed(os.path.join(GOLD, "vit_b_16.npz"), **run_vit(VT, "vit_b_16", 2))

    # (4) sort_and_drop op-level (models/rankvit.py:55-77) + whole-model rankvit
    print("rankvit")
    from peekvit.models.rankvit import RankViTBlock
    blk = RankViTBlock(num_heads=2, hidden_dim=64, mlp_dim=128, dropout=0.0, attention_dropout=0.0)
    sd_out = {}
    for N in (196, 400):
        x = torch.from_numpy(sorted_gap_tokens(2, N, 64, seed=0))
        for budget in (0.1, 0.25, 0.5, 0.75, 0.99):
            blk.set_budget(budget)
            out = blk.sort_and_drop(x)
            tok = x[:, 1:]
            idx = torch.argsort(torch.norm(tok, dim=-1), dim=-1, descending=True)   # rankvit.py:63,67
            k = out.shape[1] - 1
            sd_out[f"N{N}_b{budget}_out"] = out.numpy()
            sd_out[f"N{N}_b{budget}_idx"] = idx[:, :k].numpy()
    np.savez_compressed(os.path.join(GOLD, "sort_and_drop.npz"), **sd_out)

    rk = {}
    for name, layers, budgets in (("vit_micro", [0, 1], (0.5, 0.25)), ("vit_tiny", [1, 2, 3], (0.5,)),
                                  ("vit_b_16", [3, 6, 9], (0.5,))):
        cfg = synth.MODEL_CONFIGS[name]
        m = RVT(**cfg, rankvit_layers=layers).eval()
        synth.load_synth_weights(m, cfg, "vit", seed=0)
        x = torch.from_numpy(synth.synth_images(2, cfg["image_size"], seed=0))
        for b in budgets:
            m.set_budget(b)
            seqs, ins = [], {}
            hooks = [blk.register_forward_hook(lambda mod, i, o: seqs.append(o.shape[1])) for blk in m.encoder.layers]
            for li in layers:
                hooks.append(m.encoder.layers[li].register_forward_pre_hook(
                    lambda mod, i, li=li: ins.__setitem__(li, i[0].detach().clone())))
            with torch.no_grad():
                logits = m(x)
            for h in hooks:
                h.remove()
            rk[f"{name}_b{b}_logits"] = logits.numpy()
            rk[f"{name}_b{b}_seq"] = np.array(seqs)
            for li in layers:
                tok = ins[li][:, 1:]
                idx = torch.argsort(torch.norm(tok, dim=-1), dim=-1, descending=True)
                k = int(np.ceil(tok.shape[1] * b))
                rk[f"{name}_b{b}_keep{li}"] = idx[:, :k].numpy()
            print(f"  rank {name} layers={layers} b={b}: seq {seqs}")
        m.set_budget(1.0)
        with torch.no_grad():
            rk[f"{name}_b1.0_logits"] = m(x).numpy()
    np.savez_compressed(os.path.join(GOLD, "rankvit.npz"), **rk)

    # (5) residualvit (configs/model/residualvit_b_16.yaml settings: sigmoid gate, bias 10, temp 1, learnable token)
    print("residualvit")
    rs = {}
    # gate_bias=0 is an extra micro case whose masks contain exact zeros (relu clipping, residualvit.py:62-69)
    for tag, name, gate_bias in (("vit_micro", "vit_micro", 10), ("vit_micro_gb0", "vit_micro", 0),
                                 ("vit_b_16", "vit_b_16", 10)):
        cfg = dict(synth.MODEL_CONFIGS[name])
        extra = dict(residual_layers=["attention+mlp"] * cfg["num_layers"], gate_temp=1, add_input=False,
                     gate_type="sigmoid", gate_threshold=0.5, gate_bias=gate_bias, add_budget_token="learnable")
        torch.manual_seed(0)
        m = ResVT(**cfg, **extra).eval()
        scfg = dict(cfg, **extra)
        synth.load_synth_weights(m, scfg, "residualvit", seed=0)
        x = torch.from_numpy(synth.synth_images(2, cfg["image_size"], seed=0))
        for b in (0.2, 0.5, 1.0):
            m.set_budget(b)
            cls_rows = []
            hooks = _cls_hooks(m, cls_rows)
            with torch.no_grad():
                logits = m(x)
            for h in hooks:
                h.remove()
            masks = torch.stack([blk.mask.detach() for blk in m.encoder.layers])    # [L,B,N,1]
            rs[f"{tag}_b{b}_logits"] = logits.numpy()
            rs[f"{tag}_b{b}_masks"] = masks.numpy()
            rs[f"{tag}_b{b}_block_cls"] = torch.stack(cls_rows).numpy()
            print(f"  residual {tag} b={b}: mask mean {masks.mean():.4f} zeros {(masks == 0).float().mean():.3f}")
    np.savez_compressed(os.path.join(GOLD, "residualvit.npz"), **rs)

    # (6) error paths: capture the reference's exception types/messages (SURVEY section 8b error conventions)
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    m = VT(**cfg).eval()
    for key, fn in {
        "wrong_height": lambda: m(torch.zeros(1, 3, 40, 32)),
        "wrong_width": lambda: m(torch.zeros(1, 3, 32, 40)),
        "indivisible": lambda: VT(**dict(cfg, image_size=30)),
        "block_rank": lambda: m.encoder.layers[0](torch.zeros(4, 4)),
        "rank_registers": lambda: RVT(**cfg, num_registers=2, rankvit_layers=[0]),
        "rank_none_layers": lambda: RVT(**cfg),
        "residual_gate_type": lambda: ResVT(**cfg, gate_type="nope"),
        "residual_gumbel_threshold": lambda: ResVT(**cfg, gate_type="gumbel", gate_threshold=0.3),
        "residual_set_budget_training": lambda: ResVT(**cfg, gate_type="sigmoid").train().set_budget(0.5),
        "residual_no_budget_eval": lambda: ResVT(**cfg, gate_type="sigmoid", add_budget_token="learnable").eval()(
            torch.zeros(1, 3, 32, 32)),
    }.items():
        try:
            fn()
            meta["errors"][key] = None
        except Exception as e:  # noqa: BLE001 - recording whatever the reference raises
            meta["errors"][key] = {"type": type(e).__name__, "message": str(e)}
    # state-dict key/shape contract
    meta["state_dict"] = {}
    for tag, mod in (("vit_micro", VT(**cfg)),
                     ("rankvit_micro", RVT(**cfg, rankvit_layers=[0, 1])),
                     ("residualvit_micro", ResVT(**cfg, gate_type="sigmoid", add_budget_token="learnable"))):
        meta["state_dict"][tag] = {k: list(v.shape) for k, v in mod.state_dict().items()}
    with open(os.path.join(GOLD, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    main()
