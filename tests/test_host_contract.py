"""CPU-side tests: the C-ABI library loads and exports every symbol include/peekvit_hip.h declares, the module
surface keeps the reference's constructor / state-dict / error / attribute contract, the stock-op composite path
(CPU tensors) reproduces the reference's golden logits bit-for-bit, and the YAML `_target_` strings resolve."""
import importlib
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, rel_l2
from peekvit_amd import synth


def test_library_exports_every_declared_symbol():
    from peekvit_amd import _lib
    header = open(os.path.join(REPO, "include", "peekvit_hip.h")).read()
    declared = set(re.findall(r"\b(pv_[a-z0-9_]+)\s*\(", header)) - {"pv_gemm_args"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                                   # dlopen + getattr of every symbol (no compute call)
    assert lib.pv_version() == _lib.ABI_VERSION == 10 and lib.pv_arch() == b"gfx950" and lib.pv_operand_type() == 0
    assert b"launch" in lib.pv_error_string(-3)
    lib16 = _lib.load("f16")                            # the fp16-operand build of the same sources exports the same ABI
    assert lib16.pv_version() == 10 and lib16.pv_operand_type() == 1


def _header_gemm_fields():
    """(name, C type) of every pv_gemm_args field, parsed from include/peekvit_hip.h (comments removed)."""
    import re
    src = open(os.path.join(REPO, "include", "peekvit_hip.h")).read()
    body = src[src.index("typedef struct pv_gemm_args {"):src.index("} pv_gemm_args;")].split("{", 1)[1]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        ctype, names = decl.rsplit(" ", 1)[0], decl
        m = re.match(r"(const )?(\w+)(\*?) (.+)", decl)
        base, ptr, names = m.group(2), m.group(3), m.group(4)
        for n in names.split(","):
            n = n.strip()
            fields.append((n.lstrip("*"), "ptr" if (ptr or n.startswith("*")) else base))
    return fields


def _ctypes_kind(t):
    import ctypes as C
    return {C.c_void_p: "ptr", C.c_int64: "int64_t", C.c_uint64: "uint64_t", C.c_int32: "int32_t", C.c_float: "float"}[t]


def test_gemm_args_struct_matches_header_layout():
    import ctypes as C
    from peekvit_amd._lib import GemmArgs
    hdr = _header_gemm_fields()
    assert hdr[0] == ("struct_size", "uint64_t")          # ABI v7: the length comes first, readable whatever the caller's struct is
    assert [(n, _ctypes_kind(t)) for n, t in GemmArgs._fields_] == hdr
    assert C.sizeof(GemmArgs) == 8 + 7 * 8 + 11 * 8 + 4 + 4 + 4 * 8 + 8 + 8 + 5 * 8 + 2 * 8 + 8   # size, 7 pointers, 11 int64, qscale + epilogue, fused-LN fields, ln_eps + ksplit, colsum_partial, 5 fold pointers, range_flag, rowsq_out, res_scaled (+ padding)
    assert GemmArgs.struct_size.offset == 0 and GemmArgs.A.offset == 8
    assert GemmArgs.qscale.offset == 19 * 8 and GemmArgs.epilogue.offset == 19 * 8 + 4
    assert GemmArgs.ln_gamma.offset == 20 * 8 and GemmArgs.ln_eps.offset == 24 * 8
    assert GemmArgs.res_scaled.offset == C.sizeof(GemmArgs) - 8
    assert GemmArgs().struct_size == C.sizeof(GemmArgs) and GemmArgs(M=3).struct_size == C.sizeof(GemmArgs)


def test_integration_md_stub_matches_header_field_for_field():
    """The ctypes stub INTEGRATION.md tells a reference maintainer to paste is EXECUTED (with a stand-in for the library handle) and its
    struct compared with include/peekvit_hip.h: round 2 appended a field to the header and the library without updating the document."""
    import ctypes as C
    import re
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = doc[doc.index("class pv_gemm_args(C.Structure):"):doc.index("_lib.pv_gemm_args_size.restype")]
    ns = {"C": C}
    exec(block, ns)
    stub = ns["pv_gemm_args"]
    assert [(n, _ctypes_kind(t)) for n, t in stub._fields_] == _header_gemm_fields()
    from peekvit_amd._lib import GemmArgs, ABI_VERSION
    assert C.sizeof(stub) == C.sizeof(GemmArgs) and stub(M=5).struct_size == C.sizeof(GemmArgs)
    # the version the document asserts is the one the package binds and the library source returns
    assert int(re.search(r"_lib\.pv_version\(\) == (\d+)", doc).group(1)) == ABI_VERSION
    api = open(os.path.join(REPO, "peekvit_amd", "csrc", "pv_api.hip")).read()
    assert int(re.search(r"pv_version\(void\) \{ return (\d+); \}", api).group(1)) == ABI_VERSION


def test_library_refuses_a_struct_of_another_length():
    """No GPU needed: pv_gemm_bf16 / pv_gemm_tn_bf16 / pv_gemm_tile_rows return PV_ERR_INVALID_ARG before reading anything but the
    first field when struct_size is not the library's own sizeof(pv_gemm_args)."""
    import ctypes as C
    from peekvit_amd import _lib
    lib = _lib.load()
    assert lib.pv_version() == _lib.ABI_VERSION and lib.pv_gemm_args_size() == C.sizeof(_lib.GemmArgs)
    short = (C.c_uint64 * 1)(C.sizeof(_lib.GemmArgs) - 8)          # a one-field "struct" claiming round 2's length
    p = C.cast(short, C.POINTER(_lib.GemmArgs))
    assert lib.pv_gemm_bf16(p, None) == -1 and lib.pv_gemm_tn_bf16(p, None) == -1 and lib.pv_gemm_tile_rows(p) == -1     # PV_ERR_INVALID_ARG
    zero = (C.c_uint64 * 1)(0)
    assert lib.pv_gemm_bf16(C.cast(zero, C.POINTER(_lib.GemmArgs)), None) == -1


def test_ops_fail_loudly_without_gpu_tensors():
    from peekvit_amd import ops
    from peekvit_amd._lib import PeekvitHipError
    with pytest.raises(PeekvitHipError):
        ops.cast_bf16(torch.zeros(8))                   # CPU tensor: no silent fallback


@pytest.mark.parametrize("target", ["peekvit.models.vit.VisionTransformer", "peekvit.models.rankvit.RankVisionTransformer",
                                    "peekvit.models.residualvit.ResidualVisionTransformer"])
def test_hydra_targets_resolve(target):
    mod, cls = target.rsplit(".", 1)
    assert hasattr(importlib.import_module(mod), cls)


def _models():
    from peekvit_amd.models.vit import VisionTransformer
    from peekvit_amd.models.rankvit import RankVisionTransformer
    from peekvit_amd.models.residualvit import ResidualVisionTransformer
    return VisionTransformer, RankVisionTransformer, ResidualVisionTransformer


def test_state_dict_contract_matches_reference():
    VT, RVT, ResVT = _models()
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))["state_dict"]
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    shapes = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
    assert shapes(VT(**cfg)) == meta["vit_micro"]
    assert shapes(RVT(**cfg, rankvit_layers=[0, 1])) == meta["rankvit_micro"]
    assert shapes(ResVT(**cfg, gate_type="sigmoid", add_budget_token="learnable")) == meta["residualvit_micro"]
    # the synthetic generator covers exactly the same keys
    assert set(synth.synth_state_dict(cfg)) == set(meta["vit_micro"])


def test_error_contract_matches_reference_on_cpu():
    VT, RVT, ResVT = _models()
    err = json.load(open(os.path.join(GOLDEN, "meta.json")))["errors"]
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    m = VT(**cfg).eval()
    cases = {
        "wrong_height": lambda: m(torch.zeros(1, 3, 40, 32)),
        "wrong_width": lambda: m(torch.zeros(1, 3, 32, 40)),
        "indivisible": lambda: VT(**dict(cfg, image_size=30)),
        "block_rank": lambda: m.encoder.layers[0](torch.zeros(4, 4)),
        "rank_registers": lambda: RVT(**cfg, num_registers=2, rankvit_layers=[0]),
        "rank_none_layers": lambda: RVT(**cfg),
        "residual_gate_type": lambda: ResVT(**cfg, gate_type="nope"),
        "residual_gumbel_threshold": lambda: ResVT(**cfg, gate_type="gumbel", gate_threshold=0.3),
        "residual_set_budget_training": lambda: ResVT(**cfg, gate_type="sigmoid").train().set_budget(0.5),
        "residual_no_budget_eval": lambda: ResVT(**cfg, gate_type="sigmoid", add_budget_token="learnable").eval()(torch.zeros(1, 3, 32, 32)),
    }
    for key, fn in cases.items():
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == err[key]["type"], key
        assert str(ei.value) == err[key]["message"], key


def test_composite_path_matches_reference_golden(golden):
    """CPU tensors take the stock-op composite: it must reproduce the REAL reference's fp32 logits."""
    VT, RVT, ResVT = _models()
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    x = torch.from_numpy(synth.synth_images(2, cfg["image_size"]))
    m = VT(**cfg).eval()
    synth.load_synth_weights(m, cfg)
    with torch.no_grad():
        assert rel_l2(m(x).numpy(), golden("vit_micro")["logits"]) < 1e-6
    r = RVT(**cfg, rankvit_layers=[0, 1]).eval()
    synth.load_synth_weights(r, cfg)
    for b in (0.5, 0.25, 1.0):
        r.set_budget(b)
        with torch.no_grad():
            assert rel_l2(r(x).numpy(), golden("rankvit")[f"vit_micro_b{b}_logits"]) < 1e-6
    assert r.current_budget == 1.0 and r.encoder.layers[0].current_budget == 1.0
    extra = dict(gate_type="sigmoid", gate_bias=0, gate_temp=1, add_budget_token="learnable")
    rs = ResVT(**cfg, **extra).eval()
    synth.load_synth_weights(rs, dict(cfg, **extra), "residualvit")
    for b in (0.2, 0.5):
        rs.set_budget(b)
        with torch.no_grad():
            out = rs(x)
        assert rel_l2(out.numpy(), golden("residualvit")[f"vit_micro_gb0_b{b}_logits"]) < 1e-5
        masks = torch.stack([blk.mask for blk in rs.encoder.layers]).numpy()
        assert np.abs(masks - golden("residualvit")[f"vit_micro_gb0_b{b}_masks"]).max() < 1e-6


def test_attribute_surface_and_surgery():
    VT, RVT, ResVT = _models()
    from peekvit_amd.models.residualvit import ResidualModule
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    m = VT(**cfg, num_registers=2, num_class_tokens=2)
    for attr in ("num_classes", "patch_size", "num_class_tokens", "num_registers", "hidden_dim", "seq_length"):
        assert hasattr(m, attr)
    assert m.seq_length == 16 + 2 + 2 and isinstance(m.encoder.layers, torch.nn.Sequential)
    with torch.no_grad():
        assert m.eval()(torch.zeros(1, 3, 32, 32)).shape == (1, 10)
    m2 = VT(**cfg, remove_layers=[0])
    assert len(m2.encoder.layers) == 1
    rs = ResVT(**cfg, gate_type="sigmoid", add_budget_token="learnable")
    assert all(isinstance(b, ResidualModule) for b in rs.encoder.layers) and rs.num_budget_tokens == 1
    names = [n for n, _ in rs.named_parameters()]
    assert any("gate" in n for n in names) and any("budget" in n for n in names) and any("class" in n for n in names)


def test_adapters_match_the_reference_adapters():
    """peekvit_amd.models.adapters against the REAL reference's `adapt_torch_state_dict` / `adapt_timm_state_dict`
    (models/adapters.py:75-166) run by oracle/make_golden_aux.py on torchvision- / timm-shaped key sets: same old->new key map,
    same output order and shapes, same zero head when the label set differs (tests/golden/adapters.json)."""
    from peekvit_amd.models.adapters import adapt_timm_state_dict, adapt_torch_state_dict
    cases = json.load(open(os.path.join(GOLDEN, "adapters.json")))["cases"]
    assert len(cases) == 6
    for tag, c in cases.items():
        fn = adapt_timm_state_dict if tag.startswith("timm") else adapt_torch_state_dict
        order = [k for k, _ in c["input_shapes"]]
        sd = {k: torch.full(tuple(shape), float(i + 1)) for i, (k, shape) in enumerate(c["input_shapes"])}
        out = fn(sd, c["num_classes"])
        assert list(out.keys()) == c["order"], tag
        assert {k: list(v.shape) for k, v in out.items()} == c["shapes"], tag
        got = {}
        for nk, v in out.items():
            if float(v.abs().sum()) == 0.0:
                assert nk in c["zeroed"], (tag, nk)
            else:
                got[order[int(v.flatten()[0]) - 1]] = nk
        assert got == c["map"], tag
        assert sorted(set(out) - set(got.values())) == c["zeroed"], tag
    # the adapted keys are exactly the build's (= the reference's) state-dict keys for that architecture
    VT, _, _ = _models()
    ref_keys = set(VT(image_size=32, patch_size=16, num_layers=12, num_heads=2, hidden_dim=32, mlp_dim=64, num_classes=10).state_dict())
    assert set(cases["torch_C10"]["order"]) == ref_keys and set(cases["timm_C10"]["order"]) == ref_keys


def test_flop_model_matches_baseline_table():
    f = lambda n, **kw: synth.fwd_flops_per_image(synth.MODEL_CONFIGS[n], **kw) / 1e9
    assert abs(f("vit_tiny") - 2.800) < 2e-3 and abs(f("vit_small") - 6.171) < 2e-3 and abs(f("vit_b_16") - 35.128) < 2e-3
    seqs = [197] * 3 + [99] * 3 + [50] * 3 + [26] * 3
    assert abs(f("vit_b_16", seq_per_layer=seqs) - 16.508) < 5e-3


def test_hook_macs_equal_the_reference_hooks():
    """peekvit_amd.flops.hook_macs against the REAL reference's two counter hooks (utils/flops_count.py:27-145) fired on the REAL
    reference's models by oracle/make_golden_aux.py: per-module and total MACs, integer-equal (tests/golden/flops_hooks.json).
    Covers RankViT's shrinking sequences and ResidualViT's zero-row discounting (81 % / 92 % zero masks at gate_bias 0)."""
    from peekvit_amd import flops
    VT, RVT, ResVT = _models()
    cases = json.load(open(os.path.join(GOLDEN, "flops_hooks.json")))["cases"]
    assert len(cases) == 8
    res_extra = dict(gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5, add_budget_token="learnable")
    for tag, c in cases.items():
        cfg = dict(synth.MODEL_CONFIGS[c["config"]])
        if c["kind"] == "VisionTransformer":
            m, scfg, fam = VT(**cfg), cfg, "vit"
        elif c["kind"] == "RankVisionTransformer":
            m, scfg, fam = RVT(**cfg, rankvit_layers=[3, 6, 9] if "b_16" in tag else [0, 1]), cfg, "vit"
        else:
            extra = dict(res_extra, gate_bias=10 if "gb10" in tag else 0, residual_layers=["attention+mlp"] * cfg["num_layers"])
            m, scfg, fam = ResVT(**cfg, **extra), dict(cfg, **extra), "residualvit"
        m.eval()
        synth.load_synth_weights(m, scfg, fam, seed=0)
        if c["budget"] is not None:
            m.set_budget(c["budget"])
        r = flops.hook_macs(m, torch.from_numpy(synth.synth_images(c["batch"], cfg["image_size"], seed=0)))
        assert r["per_module_macs"] == c["per_module_macs"], tag
        assert r["total_macs"] == c["total_macs"], tag
        if "zero_rows_per_block" in c:
            assert [int((b.mask == 0).sum()) for b in m.encoder.layers] == c["zero_rows_per_block"], tag


def test_flop_accounting_conventions():
    """peekvit_amd.flops restates utils/flops_count.py: 2*MACs, bias MACs counted, zero rows discounted, RankViT shrinks S."""
    from peekvit_amd import flops
    VT, RVT, ResVT = _models()
    cfg = synth.MODEL_CONFIGS["vit_b_16"]
    m = VT(**cfg)
    f = flops.model_flops(m)
    gemm_only = synth.fwd_flops_per_image(cfg)
    assert gemm_only < f < 1.02 * gemm_only            # + biases, LayerNorms, softmax, q scaling: ~1 %
    # Linear hook arithmetic on a known case: (in*out + out) * rows, x2
    assert flops._linear_macs(10, 8, 4) == (8 * 4 + 4) * 10
    # rank: fewer tokens -> fewer flops; budget 1 -> identical
    r = RVT(**cfg, rankvit_layers=[3, 6, 9])
    seqs = [197] * 3 + [99] * 3 + [50] * 3 + [26] * 3
    assert flops.model_flops(r, seqs) < 0.5 * f and flops.model_flops(r) == f
    # residual: zero rows are discounted, soft (non-zero) masks are not
    mc = synth.MODEL_CONFIGS["vit_micro"]
    extra = dict(gate_type="sigmoid", gate_bias=0, gate_temp=1, add_budget_token="learnable")
    rs = ResVT(**mc, **extra).eval()
    synth.load_synth_weights(rs, dict(mc, **extra), "residualvit")
    rs.set_budget(0.5)
    x = torch.from_numpy(synth.synth_images(2, mc["image_size"]))
    fl, sp = flops.measured_flops(rs, x)
    assert 0.5 < sp < 1.0 and fl < flops.model_flops(rs)
    rs.set_budget(0.2)
    fl2, sp2 = flops.measured_flops(rs, x)
    assert sp2 < sp and fl2 > fl                       # golden: 81 % zeros at 0.2 vs 92 % at 0.5


def test_residualvit_training_step_matches_the_reference():
    """ResidualViT in TRAINING mode (sigmoid gates, learnable budget token, one budget per sample drawn with torch.rand(n):
    reference models/residualvit.py:541-567) - this package's module vs one step of the REAL reference model
    (tests/golden/train_step.npz): same RNG consumption, loss, every gradient norm, nine complete gradients."""
    from peekvit_amd.models.residualvit import ResidualVisionTransformer
    g = np.load(os.path.join(GOLDEN, "train_step.npz"))
    cfg = synth.MODEL_CONFIGS["vit_micro"]
    extra = dict(residual_layers=["attention+mlp"] * 2, gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5,
                 gate_bias=10, add_budget_token="learnable")
    m = ResidualVisionTransformer(**cfg, **extra)
    synth.load_synth_weights(m, dict(cfg, **extra), "residualvit", seed=0)
    m.train()
    x = torch.from_numpy(synth.synth_images(6, cfg["image_size"], seed=0))
    y = torch.arange(6) % cfg["num_classes"]
    torch.manual_seed(7)
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    name = "residualvit_micro"
    assert rel_l2(logits.detach().numpy(), g[f"{name}/logits"]) < 2e-6
    assert abs(float(loss) - float(g[f"{name}/loss"])) < 1e-6
    named = dict(m.named_parameters())
    names = [str(n) for n in g[f"{name}/names"]]
    assert names == [n for n, _ in m.named_parameters()]
    gn = np.array([float(named[n].grad.norm()) if named[n].grad is not None else 0.0 for n in names])
    assert np.allclose(gn, g[f"{name}/grad_norms"], rtol=2e-4, atol=1e-7)


def test_wgrad_slice_count_fills_whole_rounds():
    """train_engine._tn_slices: the split-K slice count of the weight-gradient GEMM is chosen so that tiles x slices workgroups fill
    whole rounds of the 256 CUs (r1's power-of-two rule left 25-44 % of the last round idle on the ViT-B/16 shapes)."""
    from peekvit_amd.train_engine import _tn_slices
    R = 2048 * 197
    for No, Ni in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        tiles = ((No + 255) // 256) * ((Ni + 255) // 256)
        s = _tn_slices(R, tiles, No, Ni)
        blocks = tiles * s
        assert 1 <= s <= 32 and R // (2 * s) >= 256
        assert blocks / (256 * -(-blocks // 256)) > 0.95, (No, Ni, s)          # >= 95 % of the last round busy
    assert _tn_slices(394, 9, 768, 768) == 1                                   # tiny batch: nothing to split


def test_workspace_size_matches_the_documented_formulas():
    """pv_workspace_size (ABI v8, SURVEY 8b's export list) is a HOST function: the scratch sizes peekvit_amd/ops.py allocates, checked here against
    the formulas written next to the entry points in include/peekvit_hip.h; bad arguments return a negative error code."""
    import ctypes as C
    from peekvit_amd import _lib
    lib = _lib.load()

    def ws(use, *dims):
        return int(lib.pv_workspace_size(use, (C.c_int64 * len(dims))(*dims), len(dims)))
    assert ws(_lib.PV_WS_COLSUM, 403456, 768) == -(-403456 // 1024) * 768 * 4 and ws(_lib.PV_WS_COLSUM, 1000, 128) == -(-1000 // 64) * 128 * 4
    assert ws(_lib.PV_WS_TRANSPOSE_COLSUM, 3000, 768, 3072) == 3072 // 64 * 768 * 4 and ws(_lib.PV_WS_TRANSPOSE_COLSUM, 70000, 8, 70656) == 69 * 8 * 4
    assert ws(_lib.PV_WS_LAYERNORM_BWD, 7, 128) == 2 * 3 * 128 * 4 and ws(_lib.PV_WS_LAYERNORM_BWD, 403456, 768) == 1024 * 3 * 768 * 4
    assert ws(_lib.PV_WS_GEMM_COLSUM_PARTIAL, 403456, 3072) == 1576 * 3072 * 4
    assert ws(_lib.PV_WS_GEMM_SPLITK, 768, 3072, 27) == 27 * 768 * 3072 * 4
    assert ws(99, 1, 1) < 0 and ws(_lib.PV_WS_COLSUM, 5) < 0 and ws(_lib.PV_WS_COLSUM, 0, 8) < 0
    assert lib.pv_workspace_size(_lib.PV_WS_COLSUM, None, 2) < 0
    from peekvit_amd import ops
    assert ops.workspace_bytes(_lib.PV_WS_LAYERNORM_BWD, 100, 384) == 25 * 3 * 384 * 4
