"""GPU parity, model level: the nn.Module surface on the HIP path against
  (a) the oracle in bf16 'same-rounding-points' mode (tight), and
  (b) golden logits captured from the REAL fp32 reference (looser: bf16 operand rounding),
plus size-independent properties at BASELINE's full batch (batch invariance, permutation equivariance)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import vit_oracle as O
from peekvit_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Tolerances (relative L2).
# TOL_BLOCK: ONE block on IDENTICAL inputs, HIP vs the oracle's same-rounding-points mode: only fp32 summation
#   order / exp ulps differ, re-rounded to bf16 a few times (measured 4e-5..7e-5 over all rows).
# TOL_E2E: whole-model logits.  bf16 MFMA operands + fp32 accumulate + fp32 residual stream is measured at
#   4.3e-3 (ViT-B/16) .. 7e-3 (vit_tiny) against the fp32 reference at random init - the same as the CPU
#   oracle's bf16 mode (4.1e-3 .. 7e-3, SURVEY 7 H1 predicted 3.6e-3..4.8e-3).  Logits depend on the CLS row only,
#   whose norm is ~15x smaller than patch rows, so bf16 rounding noise decorrelates between two bf16
#   implementations: HIP-vs-oracle(bf16) is no tighter than HIP-vs-fp32.  BASELINE's 1e-3 needs split-precision
#   operands.
TOL_BLOCK = 3e-4
TOL_E2E = 1.2e-2


def _model(kind, name, **extra):
    from peekvit_amd.models.vit import VisionTransformer
    from peekvit_amd.models.rankvit import RankVisionTransformer
    from peekvit_amd.models.residualvit import ResidualVisionTransformer
    cfg = synth.MODEL_CONFIGS[name]
    cls = dict(vit=VisionTransformer, rank=RankVisionTransformer, res=ResidualVisionTransformer)[kind]
    m = cls(**cfg, **extra)
    synth.load_synth_weights(m, dict(cfg, **extra), "residualvit" if kind == "res" else "vit", seed=0)
    return cfg, m.eval().to(DEV)


def _x(cfg, b=2):
    return torch.from_numpy(synth.synth_images(b, cfg["image_size"], seed=0))


@pytest.mark.parametrize("name", ["vit_micro", "vit_tiny", "vit_small", "vit_b_16"])
def test_vit_forward_parity(golden, name):
    from peekvit_amd import ops
    cfg, m = _model("vit", name)
    x = _x(cfg)
    n0 = ops.launch_count
    with torch.no_grad():
        logits = m(x.to(DEV)).cpu().numpy()
    assert ops.launch_count - n0 >= 4 + 7 * cfg["num_layers"], "the HIP kernels did not run"
    sd = synth.synth_state_dict(cfg)
    same = O.vit_forward(x, sd, cfg, "bf16").numpy()
    assert rel_l2(logits, same) < TOL_E2E
    assert rel_l2(logits, golden(name)["logits"]) < TOL_E2E          # vs the REAL reference's fp32 logits


def test_vit_micro_per_block_activations(golden):
    g = golden("vit_micro")
    cfg, m = _model("vit", "vit_micro")
    outs = []
    hooks = [blk.register_forward_hook(lambda mod, i, o: outs.append(o.float().cpu())) for blk in m.encoder.layers]
    with torch.no_grad():
        m(_x(cfg).to(DEV))
    for h in hooks:
        h.remove()
    for i, o in enumerate(outs):
        assert rel_l2(o.numpy(), g["block_out"][i]) < 6e-3


def test_block_level_standalone_and_surgery():
    """A block called on its own (as add_noise / remove_layers surgery relies on) runs the HIP path."""
    cfg, m = _model("vit", "vit_micro")
    x = torch.from_numpy(synth.tensor("blk/x", (2, 17, 128), "normal", seed=3, bf16=False))
    with torch.no_grad():
        y = m.encoder.layers[0](x.to(DEV)).cpu()
    ref = O.vit_block(x, synth.synth_state_dict(cfg), "encoder.layers.0.", cfg["num_heads"], 1e-5, "bf16")
    assert rel_l2(y.numpy(), ref.numpy()) < TOL_BLOCK
    m.remove_layers([1])
    assert len(m.encoder.layers) == 1
    with torch.no_grad():
        assert m(_x(cfg).to(DEV)).shape == (2, cfg["num_classes"])


@pytest.mark.parametrize("name,layers,b", [("vit_micro", [0, 1], 0.5), ("vit_micro", [0, 1], 0.25), ("vit_tiny", [1, 2, 3], 0.5),
                                           ("vit_b_16", [3, 6, 9], 0.5)])
def test_rankvit_parity(golden, name, layers, b):
    g = golden("rankvit")
    cfg, m = _model("rank", name, rankvit_layers=layers)
    x = _x(cfg)
    m.set_budget(b)
    ins = {}
    hooks = [m.encoder.layers[li].register_forward_pre_hook(lambda mod, i, li=li: ins.__setitem__(li, i[0].cpu()))
             for li in layers]
    seqs = []
    hooks += [blk.register_forward_hook(lambda mod, i, o: seqs.append(o.shape[1])) for blk in m.encoder.layers]
    with torch.no_grad():
        logits = m(x.to(DEV)).cpu().numpy()
    for h in hooks:
        h.remove()
    assert seqs == list(g[f"{name}_b{b}_seq"])                       # per-layer sequence lengths = reference
    # keep-indices: bit-exact against the oracle's ranking of the SAME block input
    for li in layers:
        _, keep = O.sort_and_drop(ins[li], b)
        assert np.array_equal(m.encoder.layers[li].last_keep.cpu().numpy().astype(np.int64), keep.numpy())
    assert rel_l2(logits, g[f"{name}_b{b}_logits"]) < 2.5e-2          # e2e indices may legitimately differ after bf16 layers
    m.set_budget(1.0)
    with torch.no_grad():
        full = m(x.to(DEV)).cpu().numpy()
    assert rel_l2(full, g[f"{name}_b1.0_logits"]) < TOL_E2E


@pytest.mark.parametrize("tag,name,gb", [("vit_micro", "vit_micro", 10), ("vit_micro_gb0", "vit_micro", 0), ("vit_b_16", "vit_b_16", 10)])
def test_residualvit_parity(golden, tag, name, gb):
    g = golden("residualvit")
    extra = dict(gate_type="sigmoid", gate_temp=1, gate_bias=gb, add_budget_token="learnable", gate_threshold=0.5)
    cfg, m = _model("res", name, **extra)
    x = _x(cfg)
    sd = synth.synth_state_dict(dict(cfg, **extra), "residualvit")
    for b in (0.2, 0.5, 1.0):
        m.set_budget(b)
        with torch.no_grad():
            logits = m(x.to(DEV)).cpu().numpy()
        masks = torch.stack([blk.mask.cpu() for blk in m.encoder.layers]).numpy()
        tr = {}
        same = O.residualvit_forward(x, sd, dict(cfg, **extra), b, "bf16", trace=tr).numpy()
        assert rel_l2(logits, same) < TOL_E2E
        assert np.abs(masks - torch.stack(tr["masks"]).numpy()).max() < 5e-3
        assert np.abs(masks[0] - g[f"{tag}_b{b}_masks"][0]).max() < 1e-5      # first block sees fp32-identical input
        assert rel_l2(logits, g[f"{tag}_b{b}_logits"]) < TOL_E2E


def test_error_contract_matches_reference():
    import json, os
    from conftest import GOLDEN
    from peekvit_amd.models.vit import VisionTransformer
    from peekvit_amd.models.rankvit import RankVisionTransformer
    from peekvit_amd.models.residualvit import ResidualVisionTransformer
    err = json.load(open(os.path.join(GOLDEN, "meta.json")))["errors"]
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    m = VisionTransformer(**cfg).eval().to(DEV)
    cases = {
        "wrong_height": lambda: m(torch.zeros(1, 3, 40, 32, device=DEV)),
        "wrong_width": lambda: m(torch.zeros(1, 3, 32, 40, device=DEV)),
        "block_rank": lambda: m.encoder.layers[0](torch.zeros(4, 4, device=DEV)),
        "rank_registers": lambda: RankVisionTransformer(**cfg, num_registers=2, rankvit_layers=[0]),
        "rank_none_layers": lambda: RankVisionTransformer(**cfg),
        "residual_no_budget_eval": lambda: ResidualVisionTransformer(**cfg, gate_type="sigmoid", add_budget_token="learnable").eval().to(DEV)(
            torch.zeros(1, 3, 32, 32, device=DEV)),
    }
    for key, fn in cases.items():
        with torch.no_grad(), pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == err[key]["type"] and str(ei.value) == err[key]["message"], key


def test_full_batch_properties_vit_b_16():
    """BASELINE config 3 size (B=2048, 224x224): batch invariance + permutation equivariance, bit-exact."""
    cfg, m = _model("vit", "vit_b_16")
    gen = torch.Generator(device="cpu").manual_seed(0)
    small = torch.randn(4, 3, 224, 224, generator=gen).to(torch.bfloat16).float()
    B = 2048
    big = torch.randn(B, 3, 224, 224, generator=gen).to(torch.bfloat16).float()
    pos = [0, 777, 1500, 2047]
    for p, s in zip(pos, small):
        big[p] = s
    with torch.no_grad():
        ls = m(small.to(DEV)).cpu()
        lb = m(big.to(DEV)).cpu()
    assert torch.isfinite(lb).all()
    assert torch.equal(lb[pos], ls)                                    # an image's logits do not depend on its batch
    perm = torch.randperm(B, generator=gen)
    with torch.no_grad():
        lp = m(big[perm].to(DEV)).cpu()
    assert torch.equal(lp, lb[perm])                                   # permuting images permutes logits


@pytest.mark.parametrize("name", ["vit_tiny", "vit_small", "vit_b_16"])
def test_every_block_on_identical_inputs(name):
    """Each encoder block of the real configs, fed the ORACLE's input, against the oracle's same-rounding-points
    output: isolates kernel correctness from the chaotic e2e accumulation (all rows, and the CLS row alone)."""
    cfg, m = _model("vit", name)
    sd = synth.synth_state_dict(cfg)
    x = _x(cfg)
    t = O.embed_tokens(x, sd, cfg, "bf16") + torch.from_numpy(sd["encoder.pos_embedding"])
    with torch.no_grad():
        from peekvit_amd import engine
        assert rel_l2(engine.embed_tokens(m, x.to(DEV)).cpu().numpy(), t.numpy()) < 1e-6
        for i, blk in enumerate(m.encoder.layers):
            ref = O.vit_block(t, sd, f"encoder.layers.{i}.", cfg["num_heads"], 1e-5, "bf16")
            got = blk(t.to(DEV)).cpu()
            assert rel_l2(got.numpy(), ref.numpy()) < TOL_BLOCK, i
            t = ref


def test_hip_graph_replay_matches_eager():
    """The forward is capturable into a hipGraph (stateless, allocation-free C ABI) and replays bit-identically."""
    from peekvit_amd.graph import GraphedForward
    cfg, m = _model("vit", "vit_tiny")
    x = _x(cfg, 4).to(DEV)
    with torch.no_grad():
        eager = m(x).clone()
    g = GraphedForward(m, x)
    assert torch.equal(g(x), eager)
    x2 = torch.flip(x, dims=[0])
    assert torch.equal(g(x2), torch.flip(eager, dims=[0]))


def test_uint8_nhwc_input_is_bit_identical_to_normalised_fp32_nchw():
    """raw uint8 NHWC images through the fused gather == ToTensor+Normalize on the host, then the fp32 path."""
    from peekvit_amd.engine import IMAGENET_MEAN, IMAGENET_STD
    cfg, m = _model("vit", "vit_tiny")
    gen = torch.Generator().manual_seed(3)
    raw = torch.randint(0, 256, (3, cfg["image_size"], cfg["image_size"], 3), generator=gen, dtype=torch.uint8)
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    x = (raw.permute(0, 3, 1, 2).float().div(255.0) - mean) / std          # torchvision ToTensor + Normalize
    with torch.no_grad():
        a = m(x.contiguous().to(DEV)).cpu()
        b = m(raw.to(DEV)).cpu()
    assert torch.equal(a, b)


def test_vit_384_long_sequence_forward():
    """384x384 at patch 16 -> S = 577 (> the 416 tokens the LDS-resident attention holds): whole forward vs the stock-op composite."""
    from peekvit_amd.models.vit import VisionTransformer
    torch.manual_seed(0)
    m = VisionTransformer(image_size=384, patch_size=16, num_layers=2, num_heads=2, hidden_dim=128, mlp_dim=256, num_classes=10)
    torch.nn.init.normal_(m.head.weight, std=0.05)
    m = m.eval().to("cuda:0")
    x = torch.randn(3, 3, 384, 384, device="cuda:0")
    with torch.no_grad():
        got = m(x)
        ref = m._composite_head(m.encoder(m._composite_tokens(x)))
    assert rel_l2(got.cpu(), ref.cpu()) < 1.2e-2


def test_forward_leaves_the_module_tree_untouched():
    """The engine's per-block hints must never register modules: state_dict keys / named_parameters are the reference's before and
    after a forward on the HIP path (checkpoints written after evaluation must load in a peekvit checkout)."""
    from peekvit_amd import synth
    from peekvit_amd.models.vit import VisionTransformer
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    m = VisionTransformer(**cfg)
    synth.load_synth_weights(m, cfg)
    m = m.eval().to("cuda:0")
    keys, names = list(m.state_dict().keys()), [n for n, _ in m.named_parameters()]
    with torch.no_grad():
        m(torch.randn(2, 3, cfg["image_size"], cfg["image_size"], device="cuda:0"))
    assert list(m.state_dict().keys()) == keys and [n for n, _ in m.named_parameters()] == names
    assert not any("_pv" in k for k in keys)


def test_layernorm_folding_opt_in(monkeypatch):
    """PEEKVIT_AMD_FOLD_LN=1: same model, LayerNorm folded into the producer / consumer GEMM epilogues (no LayerNorm launch after the
    first block): logits agree with the default path to the operand-rounding noise, and the fold path really ran."""
    from peekvit_amd import engine, ops, synth
    from peekvit_amd.models.vit import VisionTransformer
    cfg = synth.MODEL_CONFIGS["vit_small"]
    m = VisionTransformer(**cfg)
    synth.load_synth_weights(m, cfg)
    m = m.eval().to("cuda:0")
    x = torch.from_numpy(synth.synth_images(12, cfg["image_size"], seed=0)).to("cuda:0")      # 12 x 197 = 2364 rows >= 2048
    with torch.no_grad():
        ref = m(x)
        monkeypatch.setattr(engine, "_FOLD_LN", True)
        with ops.KernelTimer() as kt:
            got = m(x)
    ks = kt.summary()
    assert "pv_rowstat_finalize" in ks and ks["pv_layernorm_bf16"]["launches"] == 1          # only block 0's ln_1 is a LayerNorm launch
    assert rel_l2(got.cpu(), ref.cpu()) < 1.2e-2
