"""GPU parity, model level: the nn.Module surface on the HIP path against
  (a) the oracle in bf16 'same-rounding-points' mode (tight), and
  (b) golden logits captured from the REAL fp32 reference (looser: bf16 operand rounding),
plus size-independent properties at BASELINE's full batch (batch invariance, permutation equivariance)."""
import warnings

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import vit_oracle as O
from peekvit_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# The DEFAULT path is what these tests run: precision mode "auto" = fp16 operands behind the range guard (engine.py).
# Tolerances (relative L2).
# TOL_CONTRACT: BASELINE.json north_star - logits within 1e-3 relative of the reference's fp32 forward, asserted against the golden
#   vectors captured from the REAL reference (measured 5.4e-4 .. 6.7e-4 with fp16 operands).
# TOL_BLOCK: ONE block on IDENTICAL inputs, HIP vs the oracle's same-rounding-points mode: only fp32 summation
#   order / exp ulps differ, re-rounded to 16 bits a few times (measured 4e-5..7e-5 over all rows).
# TOL_SAME: whole-model logits vs the oracle restatement with the SAME operand rounding points: with fp16 operands the rounding noise
#   (~6e-4) is of the order of the implementation differences (fp32 accumulation order, exp2 softmax, table GELU).
# TOL_BF16: the explicit "bf16" mode (4.3e-3 ViT-B/16 .. 7e-3 vit_tiny against the fp32 reference: 8-bit operand mantissas).
TOL_CONTRACT = 1e-3
TOL_BLOCK = 3e-4
TOL_SAME = 1.5e-3
TOL_BF16 = 1.2e-2


def _op():
    """Operand type of the default inference path ("f16" in mode auto): the oracle mode with the same rounding points."""
    from peekvit_amd import engine
    return engine.inference_operand()


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
    same = O.vit_forward(x, sd, cfg, _op()).numpy()
    assert rel_l2(logits, same) < TOL_SAME
    assert rel_l2(logits, golden(name)["logits"]) < TOL_CONTRACT     # vs the REAL reference's fp32 logits: the north_star tolerance
    from peekvit_amd import engine
    with torch.no_grad(), engine.precision("bf16"):                  # the explicit bf16-operand mode stays available
        lb = m(x.to(DEV)).cpu().numpy()
    assert rel_l2(lb, golden(name)["logits"]) < TOL_BF16 and rel_l2(lb, O.vit_forward(x, sd, cfg, "bf16").numpy()) < TOL_BF16


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
        assert rel_l2(o.numpy(), g["block_out"][i]) < TOL_CONTRACT


def test_block_level_standalone_and_surgery():
    """A block called on its own (as add_noise / remove_layers surgery relies on) runs the HIP path."""
    cfg, m = _model("vit", "vit_micro")
    x = torch.from_numpy(synth.tensor("blk/x", (2, 17, 128), "normal", seed=3, bf16=False))
    with torch.no_grad():
        y = m.encoder.layers[0](x.to(DEV)).cpu()
    ref = O.vit_block(x, synth.synth_state_dict(cfg), "encoder.layers.0.", cfg["num_heads"], 1e-5, _op())
    assert rel_l2(y.numpy(), ref.numpy()) < TOL_BLOCK
    m.remove_layers([1])
    assert len(m.encoder.layers) == 1
    with torch.no_grad():
        assert m(_x(cfg).to(DEV)).shape == (2, cfg["num_classes"])


def test_noise_block_splice_keeps_the_hip_path_running():
    """validate/test.py:72-111 + utils/utils.py:162-191: the evaluation harness splices a NoiseBlock (stock PyTorch, RNG-driven) into
    encoder.layers and sweeps its value.  The blocks around it keep running on the MI355X kernels; the block in front of it fuses
    nothing for a consumer that is not a plain block; the logits equal the CPU composite's for the same dropped token positions."""
    from peekvit_amd import ops
    from peekvit_amd.harness.noise import add_noise
    from peekvit_amd.models.vit import VisionTransformer
    cfg, m = _model("vit", "vit_tiny")
    x = _x(cfg, 4)
    with torch.no_grad():
        clean = m(x.to(DEV)).cpu().numpy()
    nm = add_noise(m, layer=2, noise_type="token_drop", prob=0.25)
    ref = VisionTransformer(**cfg)
    synth.load_synth_weights(ref, cfg, "vit", seed=0)
    ref = ref.eval()
    rn = add_noise(ref, layer=2, noise_type="token_drop", prob=0.25)
    n0 = ops.launch_count
    with torch.no_grad():
        torch.manual_seed(11); got = m(x.to(DEV)).cpu().numpy()          # torch.randperm draws the positions from the HOST generator
        torch.manual_seed(11); want = ref(x).numpy()
    assert ops.launch_count - n0 >= 4 + 6 * cfg["num_layers"], "the blocks around the NoiseBlock left the HIP path"   # (30 observed: 4 blocks x 6-7 + stem + head)
    assert rel_l2(got, want) < TOL_CONTRACT and rel_l2(got, clean) > 5e-2       # same noise, and it did something
    nm.set_value(0.0); rn.set_value(0.0)
    with torch.no_grad():
        assert rel_l2(m(x.to(DEV)).cpu().numpy(), clean) < 1e-4                # prob 0: the clean channel (other summation order at most)
    # gaussian noise at a huge SNR is the clean channel too; at -10 dB it is not
    del m.encoder.layers[2]
    gm = add_noise(m, layer=1, noise_type="gaussian")
    with torch.no_grad():
        gm.set_value(200.0); quiet = m(x.to(DEV)).cpu().numpy()
        gm.set_value(-10.0); loud = m(x.to(DEV)).cpu().numpy()
    assert rel_l2(quiet, clean) < TOL_CONTRACT and rel_l2(loud, clean) > 5e-2 and np.isfinite(loud).all()


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
    # END-TO-END against the REAL reference.  What happens on an MI355X for these golden inputs is on record
    # (scripts/parity_observed.py -> profiles/r03_parity_observed.json): in ALL four cases every ranked layer keeps exactly the
    # reference's token SET (the 16-bit layers in front of a ranking never resolve a near-tie the other way here) and the logits are
    # 5.8e-4 .. 7.5e-4 from the reference's - so both are asserted unconditionally; round 2 chose between 1e-3 and 2.5e-2 by a branch
    # that only a dropped print recorded.
    for li in layers:
        assert np.array_equal(np.sort(m.encoder.layers[li].last_keep.cpu().numpy().astype(np.int64), axis=1),
                              np.sort(g[f"{name}_b{b}_keep{li}"], axis=1)), f"layer {li}: kept set differs from the reference's"
    assert rel_l2(logits, g[f"{name}_b{b}_logits"]) < TOL_CONTRACT
    m.set_budget(1.0)
    with torch.no_grad():
        full = m(x.to(DEV)).cpu().numpy()
    assert rel_l2(full, g[f"{name}_b1.0_logits"]) < TOL_CONTRACT


def test_rankvit_keep_sets_do_not_depend_on_the_batch_size():
    """Round 2 ADVICE: at large batches the token norms a ranked block sorts by come out of the previous block's fc2 epilogue (per-tile
    sums of squares, another summation order than pv_token_norm's) and LayerNorm is folded into the GEMMs - near-tied tokens could be
    kept or dropped differently depending on the batch an image arrives in.  RankViT-B/16 [3,6,9] @ 0.5: the two golden images inside
    a batch of 64 (256-row tile kernels, fused norms, folded LayerNorm) keep exactly the token sets they keep as a batch of 2, which are
    the reference's (test_rankvit_parity)."""
    cfg, m = _model("rank", "vit_b_16", rankvit_layers=[3, 6, 9])
    m.set_budget(0.5)
    small = _x(cfg)
    big = torch.from_numpy(synth.synth_images(64, cfg["image_size"], seed=5))
    big[7], big[40] = small[0], small[1]
    keeps = {}
    for tag, x in (("small", small), ("big", big)):
        with torch.no_grad():
            logits = m(x.to(DEV)).cpu()
        keeps[tag] = ([np.sort(m.encoder.layers[li].last_keep.cpu().numpy().astype(np.int64), axis=1) for li in (3, 6, 9)], logits)
    for a, b in zip(keeps["small"][0], keeps["big"][0]):
        assert np.array_equal(a, b[[7, 40]])
    assert rel_l2(keeps["big"][1][[7, 40]].numpy(), keeps["small"][1].numpy()) < TOL_CONTRACT


def test_full_batch_properties_rankvit_b_16(golden):
    """BASELINE config 4 size (RankViT-B/16 [3,6,9] @ 0.5, B = 2048): the two golden images, wherever they sit in the big batch, keep the
    token sets the REFERENCE keeps for them at every ranked layer and answer within the contract; permuting the batch permutes the kept
    indices and the logits bit for bit (ranking, compaction and every GEMM are per-image deterministic at a fixed batch size); every kept
    index is an image token and no token is kept twice."""
    g = golden("rankvit")
    cfg, m = _model("rank", "vit_b_16", rankvit_layers=[3, 6, 9])
    m.set_budget(0.5)
    small = _x(cfg)
    gen = torch.Generator(device="cpu").manual_seed(9)
    B = 2048
    big = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=gen).to(torch.bfloat16).float()
    pos = [3, 1999]
    big[pos[0]], big[pos[1]] = small[0], small[1]
    with torch.no_grad():
        lb = m(big.to(DEV)).cpu()
    keeps = [m.encoder.layers[li].last_keep.cpu().clone() for li in (3, 6, 9)]
    assert torch.isfinite(lb).all()
    S = 197
    for li, k in zip((3, 6, 9), keeps):
        n_in = S - 1
        S = 1 + -(-n_in // 2)
        assert k.shape == (B, S - 1) and int(k.min()) >= 0 and int(k.max()) < n_in
        ks = np.sort(k.numpy().astype(np.int64), axis=1)
        assert (np.diff(ks, axis=1) > 0).all()                           # no token twice
        ref = np.sort(g[f"vit_b_16_b0.5_keep{li}"].astype(np.int64), axis=1)
        assert np.array_equal(ks[pos], ref)                              # = the reference's kept sets (tests/golden/rankvit.npz)
    assert rel_l2(lb[pos].numpy(), g["vit_b_16_b0.5_logits"]) < TOL_CONTRACT
    perm = torch.randperm(B, generator=gen)
    with torch.no_grad():
        lp = m(big[perm].to(DEV)).cpu()
    assert torch.equal(lp, lb[perm])
    for li, k in zip((3, 6, 9), keeps):
        assert torch.equal(m.encoder.layers[li].last_keep.cpu(), k[perm])


@pytest.mark.parametrize("tag,name,gb", [("vit_micro", "vit_micro", 10), ("vit_micro_gb0", "vit_micro", 0), ("vit_b_16", "vit_b_16", 10)])
def test_residualvit_parity(golden, tag, name, gb):
    g = golden("residualvit")
    extra = dict(gate_type="sigmoid", gate_temp=1, gate_bias=gb, add_budget_token="learnable", gate_threshold=0.5)
    cfg, m = _model("res", name, **extra)
    x = _x(cfg)
    sd = synth.synth_state_dict(dict(cfg, **extra), "residualvit")
    from peekvit_amd import engine
    for b in (0.2, 0.5, 1.0):
        m.set_budget(b)
        t0, f0 = engine.selfcheck_trips, engine.fallback_count
        with torch.no_grad():
            logits = m(x.to(DEV)).cpu().numpy()
        # mode "auto" measures every new (parameters, budget, batch size) against its split-operand arithmetic on the first forward
        # (engine.run_guarded, "contract self-check"); a trip means THIS forward already came from the bf16x3 mode
        tripped = engine.selfcheck_trips > t0
        mlp_split = engine.guard_state(m).mlp_hybrid          # (round 6: the self-check's first escalation step - the MLP halves in split precision, still no whole-forward fallback)
        assert engine.fallback_count - f0 == (1 if tripped else 0)
        masks = torch.stack([blk.mask.cpu() for blk in m.encoder.layers]).numpy()
        assert masks.shape[1] == x.shape[0]                                       # the probe's slice never overwrites what the blocks remember
        thr = torch.stack([blk.residual_gate.threshold.cpu() for blk in m.encoder.layers])      # left behind like ResidualGate.forward does (utils.py:131)
        assert thr.shape == (cfg["num_layers"], x.shape[0], 1, 1) and bool(((thr > 0) & (thr < 1)).all())
        tr = {}
        same = O.residualvit_forward(x, sd, dict(cfg, **extra), b, "fp32" if (tripped or mlp_split) else _op(), trace=tr).numpy()
        assert rel_l2(logits, same) < (1e-4 if tripped else 9e-4 if mlp_split else TOL_SAME)
        assert np.abs(masks - torch.stack(tr["masks"]).numpy()).max() < 5e-3
        if "thresholds" in tr:
            assert np.abs(thr.view(cfg["num_layers"], -1).numpy() - torch.stack(tr["thresholds"]).view(cfg["num_layers"], -1).numpy()).max() < 5e-3
        assert np.abs(masks[0] - g[f"{tag}_b{b}_masks"][0]).max() < 1e-5      # first block sees fp32-identical input
        assert np.abs(masks - g[f"{tag}_b{b}_masks"]).max() < 2e-3            # every block's mask vs the REAL reference's
        if np.linalg.norm(g[f"{tag}_b{b}_logits"]) > 0:
            # the contract tolerance on EVERY case.  Round 3 allowed 1.25e-3 on one of them: the 2-layer, 18-token, width-128 toy with the
            # reference's gate bias of 10 at budget 0.2 measures 1.07e-3 on plain fp16 operands with no guard bit raised (the CPU oracle with
            # the same rounding points: 1.2e-3 - thirteen rounding sites of 1 - 4.6e-4 each on a model too small to average them out; its
            # masks are 0.05 .. 0.44, nowhere near the gate threshold).  The self-check sees exactly that and answers from bf16x3.
            assert rel_l2(logits, g[f"{tag}_b{b}_logits"]) < TOL_CONTRACT
            if (tag, b) == ("vit_micro", 0.2):
                # round 6: the self-check's FIRST step (MLP halves in split precision) brings the toy inside the limit - no whole-forward fallback any more
                assert (mlp_split and not tripped and rel_l2(logits, g[f"{tag}_b{b}_logits"]) < 9e-4) or (tripped and rel_l2(logits, g[f"{tag}_b{b}_logits"]) < 1e-4)
        with torch.no_grad():                                                    # the verdict is kept: no second probe, same arithmetic again
            c0 = engine.selfcheck_count
            again = m(x.to(DEV)).cpu().numpy()
        assert engine.selfcheck_count == c0 and np.array_equal(again, logits)


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


def test_full_batch_properties_vit_b_16(monkeypatch):
    """BASELINE config 3 size (B=2048, 224x224): permutation equivariance bit-exact on the default path; batch invariance
    bit-exact with LayerNorm folding off (one arithmetic for every batch size) and within the contract tolerance with the default
    (a 2048-image batch folds LayerNorm into its GEMMs, a 4-image batch does not: engine._FOLD_LN)."""
    from peekvit_amd import engine
    cfg, m = _model("vit", "vit_b_16")
    gen = torch.Generator(device="cpu").manual_seed(0)
    small = torch.randn(4, 3, 224, 224, generator=gen).to(torch.bfloat16).float()
    B = 2048
    big = torch.randn(B, 3, 224, 224, generator=gen).to(torch.bfloat16).float()
    pos = [0, 777, 1500, 2047]
    for p, s in zip(pos, small):
        big[p] = s
    big_d = big.to(DEV)
    with torch.no_grad():
        ls = m(small.to(DEV)).cpu()
        lb = m(big_d).cpu()
    assert torch.isfinite(lb).all()
    assert rel_l2(lb[pos], ls) < TOL_CONTRACT                          # default path: folded vs unfolded LayerNorm arithmetic
    perm = torch.randperm(B, generator=gen)
    with torch.no_grad():
        lp = m(big[perm].to(DEV)).cpu()
    assert torch.equal(lp, lb[perm])                                   # permuting images permutes logits, bit for bit
    # with the two batch-size dependent forms off - LayerNorm folding (large batches) and split-K residual GEMMs (small batches) - there is
    # one arithmetic for every batch size: an image's logits do not depend on the batch it travels in, bit for bit
    monkeypatch.setattr(engine, "_FOLD_LN", False)
    monkeypatch.setattr(engine, "_SMALL_M_SPLITK", False)
    with torch.no_grad():
        lb0 = m(big_d).cpu()
        ls0 = m(small.to(DEV)).cpu()
    assert torch.equal(lb0[pos], ls0)
    assert rel_l2(ls0, ls) < TOL_CONTRACT       # split-K only reorders fp32 sums, but a last-bit change upstream flips 16-bit roundings downstream


def test_full_batch_properties_vit_small(monkeypatch):
    """BASELINE config 2 size (vit_small, B = 512, 224x224: 788 row tiles of the full-row GEMM = three rounds of 128-row tiles + one of
    64-row tiles on 256 CUs, the deep-pipelined K loop at K = 384 and 1536): permutation equivariance bit-exact; an image's logits do not
    depend on the batch it travels in, bit for bit, once the small-batch split-K form is off; the contract against the CPU oracle on a
    sample of the big batch."""
    from peekvit_amd import engine
    cfg, m = _model("vit", "vit_small")
    gen = torch.Generator(device="cpu").manual_seed(1)
    small = torch.randn(8, 3, 224, 224, generator=gen).to(torch.bfloat16).float()
    B = 512
    big = torch.randn(B, 3, 224, 224, generator=gen).to(torch.bfloat16).float()
    pos = [0, 63, 64, 200, 255, 256, 400, 511]
    for p, s in zip(pos, small):
        big[p] = s
    with torch.no_grad():
        lb = m(big.to(DEV)).cpu()
        ls = m(small.to(DEV)).cpu()
    assert torch.isfinite(lb).all()
    assert rel_l2(lb[pos], ls) < TOL_CONTRACT
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg).items()}
    with torch.no_grad():
        ref = O.vit_forward(small, sd, cfg, "fp32")
    assert rel_l2(lb[pos], ref) < TOL_CONTRACT                         # the big batch against the reference's arithmetic
    perm = torch.randperm(B, generator=gen)
    with torch.no_grad():
        lp = m(big[perm].to(DEV)).cpu()
    assert torch.equal(lp, lb[perm])
    monkeypatch.setattr(engine, "_SMALL_M_SPLITK", False)
    with torch.no_grad():
        ls0 = m(small.to(DEV)).cpu()
        lb0 = m(big.to(DEV)).cpu()
    assert torch.equal(lb0[pos], ls0)


def test_default_path_with_fused_layernorm_meets_the_contract():
    """At larger batches the default path has ONE all-token LayerNorm launch (block 0's ln_1; plus the last block's two launches on its
    class-token rows): ViT-B/16 folds the others into its GEMMs
    (pv_rowstat_finalize runs), the narrow models compute them in the full-row GEMM epilogue.  Logits against the fp32 CPU oracle (= the
    reference's arithmetic) at BASELINE's 1e-3."""
    from peekvit_amd import ops
    for name, B, folds in (("vit_b_16", 64, True), ("vit_small", 96, False), ("vit_tiny", 96, False)):
        cfg, m = _model("vit", name)
        x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(11))
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg).items()}
        with torch.no_grad(), ops.KernelTimer() as kt:
            got = m(x.to(DEV)).float().cpu()
        torch.cuda.synchronize()
        ks = kt.summary()
        assert ("pv_rowstat_finalize" in ks) == folds and ks["pv_layernorm_bf16"]["launches"] == 1 + 2, (name, sorted(ks))
        torch.set_num_threads(16)
        with torch.no_grad():
            ref = O.vit_forward(x, sd, cfg, "fp32")
        err = rel_l2(got, ref)
        print(f"{name} B={B} {'folded' if folds else 'epilogue-fused'} LayerNorm: logits rel-L2 {err:.2e}")
        assert err < TOL_CONTRACT, (name, err)


@pytest.mark.parametrize("name", ["vit_tiny", "vit_small", "vit_b_16"])
def test_every_block_on_identical_inputs(name):
    """Each encoder block of the real configs, fed the ORACLE's input, against the oracle's same-rounding-points
    output: isolates kernel correctness from the chaotic e2e accumulation (all rows, and the CLS row alone)."""
    cfg, m = _model("vit", name)
    sd = synth.synth_state_dict(cfg)
    x = _x(cfg)
    t = O.embed_tokens(x, sd, cfg, _op()) + torch.from_numpy(sd["encoder.pos_embedding"])
    with torch.no_grad():
        from peekvit_amd import engine
        assert rel_l2(engine.embed_tokens(m, x.to(DEV)).cpu().numpy(), t.numpy()) < 1e-6
        for i, blk in enumerate(m.encoder.layers):
            ref = O.vit_block(t, sd, f"encoder.layers.{i}.", cfg["num_heads"], 1e-5, _op())
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
    from peekvit_amd import engine
    with torch.no_grad(), engine.precision("f16"):                        # the same arithmetic on both inputs: bit-identical
        a = m(x.contiguous().to(DEV)).cpu()
        b = m(raw.to(DEV)).cpu()
    assert torch.equal(a, b)
    with torch.no_grad():                                                  # mode "auto" (whatever its self-check decides for these noise images)
        a = m(x.contiguous().to(DEV)).cpu()
        b = m(raw.to(DEV)).cpu()
    assert rel_l2(b.numpy(), a.numpy()) < 1e-5


def test_graph_capture_of_uint8_input_in_the_split_operand_mode():
    """Round-4 review: a GraphedForward on uint8 NHWC input whose model runs in bf16x3 (a verdict of the self-check, a sticky guard) normalises
    the image with device constants - building those per forward was a pageable host-to-device copy, which stream capture refuses."""
    from peekvit_amd import engine
    from peekvit_amd.graph import GraphedForward
    cfg, m = _model("vit", "vit_tiny")
    raw = torch.randint(0, 256, (4, cfg["image_size"], cfg["image_size"], 3), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).to(DEV)
    engine._normcache.clear()
    with torch.no_grad(), engine.precision("bf16x3"):
        g = GraphedForward(m, raw)
        y = g(raw).clone()
        assert torch.equal(y, m(raw)) and torch.isfinite(y).all()
        raw2 = torch.flip(raw, dims=[0])
        assert torch.equal(g(raw2), m(raw2))


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
    # ... and against the CPU ORACLE (the reference's arithmetic, pinned by the golden vectors) at the contract's tolerance (round 3: the
    # module's own stock-op composite is pinned to the reference at micro size only)
    cfg = dict(image_size=384, patch_size=16, num_layers=2, num_heads=2, hidden_dim=128, mlp_dim=256, num_classes=10)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = O.vit_forward(x.cpu(), sd, cfg, "fp32")
    assert rel_l2(got.cpu(), want) < TOL_CONTRACT


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


def test_layernorm_folding_on_off(monkeypatch):
    """Same model with LayerNorm folded into the producer / consumer GEMM epilogues (default at this batch) and with folding off:
    logits agree to the operand-rounding noise, and each path really ran."""
    from peekvit_amd import engine, ops, synth
    from peekvit_amd.models.vit import VisionTransformer
    cfg = synth.MODEL_CONFIGS["vit_b_16"]
    m = VisionTransformer(**cfg)
    synth.load_synth_weights(m, cfg)
    m = m.eval().to("cuda:0")
    x = torch.from_numpy(synth.synth_images(56, cfg["image_size"], seed=0)).to("cuda:0")      # 56 x 197 rows: every token GEMM on 256-row tiles
    with torch.no_grad():
        with ops.KernelTimer() as kt:
            got = m(x)
        monkeypatch.setattr(engine, "_FOLD_LN", False)
        with ops.KernelTimer() as kt0:
            ref = m(x)
    torch.cuda.synchronize()
    ks, ks0 = kt.summary(), kt0.summary()
    # all-token LayerNorm launches: block 0's ln_1 only; the last block normalises its class-token rows in two small launches more
    assert "pv_rowstat_finalize" in ks and ks["pv_layernorm_bf16"]["launches"] == 1 + 2
    assert "pv_rowstat_finalize" not in ks0 and ks0["pv_layernorm_bf16"]["launches"] == 2 * cfg["num_layers"] + 1
    assert rel_l2(got.cpu(), ref.cpu()) < 2e-3


def test_residual_gate_emits_the_first_layernorm(monkeypatch):
    """ResidualViT: the gate kernel holds every token row in registers and also emits row_scale * LN1(masked row), the first step of the
    masked block - bit-identical to the separate LayerNorm launch it replaces (same row arithmetic), one pass over the tokens less per block."""
    from peekvit_amd import engine, ops
    extra = dict(gate_type="sigmoid", gate_temp=1, gate_bias=10, add_budget_token="learnable", gate_threshold=0.5)
    for name, batch in (("vit_b_16", 9), ("vit_micro", 5)):
        cfg, m = _model("res", name, **extra)
        m.set_budget(0.5)
        x = _x(cfg, batch).to(DEV)
        with torch.no_grad():
            with ops.KernelTimer() as kt:
                got = m(x)
            monkeypatch.setattr(engine, "_GATE_LN1", False)
            with ops.KernelTimer() as kt0:
                ref = m(x)
            monkeypatch.setattr(engine, "_GATE_LN1", True)
        torch.cuda.synchronize()
        L = cfg["num_layers"]
        # all-token LayerNorm launches: LN2 of the L - 1 full blocks (+ the last block's two class-row launches) vs LN1 and LN2 of every block
        assert kt.summary()["pv_layernorm_bf16"]["launches"] == (L - 1) + 2 and kt0.summary()["pv_layernorm_bf16"]["launches"] == (2 * L - 1) + 2
        assert torch.equal(got, ref)


@pytest.mark.parametrize("kind,name,batch,extra", [("vit", "vit_b_16", 56, {}), ("vit", "vit_tiny", 5, {}), ("vit", "vit_micro", 3, {"num_class_tokens": 2}),
                                                   ("rank", "vit_b_16", 56, {"rankvit_layers": [3, 6, 9]}),
                                                   ("rank", "vit_tiny", 4, {"rankvit_layers": [1, 3]}),
                                                   ("res", "vit_b_16", 9, dict(gate_type="sigmoid", gate_temp=1, gate_bias=10, add_budget_token="learnable",
                                                                               gate_threshold=0.5))])
def test_last_block_computes_class_token_rows_only(monkeypatch, kind, name, batch, extra):
    """A model forward reads only the class-token rows of the last block's output (models/vit.py:242-246), so that block computes k | v for
    every token but q, out-proj and the MLP for the class-token rows alone (engine.block_forward_rows).  Same logits as the all-rows
    block to the operand-rounding noise; a forward hook on the block (someone looks at its output) switches the shortcut off."""
    from peekvit_amd import engine, ops
    cfg, m = _model(kind, name, **extra)
    if kind == "rank" or extra.get("add_budget_token"):
        m.set_budget(0.5)
    x = _x(cfg, batch).to(DEV)
    with torch.no_grad():
        with ops.KernelTimer() as kt:
            got = m(x)
        masks = [blk.mask.clone() for blk in m.encoder.layers if getattr(blk, "mask", None) is not None]
        seen = []
        h = m.encoder.layers[-1].register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
        with ops.KernelTimer() as kth:
            hooked = m(x)
        h.remove()
        monkeypatch.setattr(engine, "_LAST_BLOCK_ROWS", False)
        with ops.KernelTimer() as kt0:
            ref = m(x)
    torch.cuda.synchronize()
    assert kt.summary()["pv_attention_rows_bf16"]["launches"] == 1 and kt.summary()["pv_attention_bf16"]["launches"] == cfg["num_layers"] - 1
    assert "pv_attention_rows_bf16" not in kt0.summary() and kt0.summary()["pv_attention_bf16"]["launches"] == cfg["num_layers"]
    assert "pv_attention_rows_bf16" not in kth.summary() and len(seen) == 1 and seen[0][0] == batch and seen[0][1] > extra.get("num_class_tokens", 1)
    assert torch.equal(hooked, ref)
    assert rel_l2(got.cpu(), ref.cpu()) < 5e-4
    masks0 = [blk.mask for blk in m.encoder.layers if getattr(blk, "mask", None) is not None]
    assert len(masks) == len(masks0) and all(torch.equal(a, b) for a, b in zip(masks, masks0))      # ResidualViT: the full masks either way


def test_rank_norms_come_from_the_fc2_epilogue(monkeypatch):
    """RankViT: the token norms of a ranked block are left behind by the previous block's fc2 epilogue (pv_gemm_args.rowsq_out ->
    pv_rank_topk_partials): no pv_token_norm pass, the same kept tokens as the standalone norm kernel, bit-identical block outputs."""
    from peekvit_amd import engine, ops
    cfg, m = _model("rank", "vit_b_16", rankvit_layers=[3, 6, 9])
    m.set_budget(0.5)
    # 224 images: even the fc2 before layer 9 (224 x 50 rows x 3 column tiles = 132 tiles) runs on the 256-row tile kernel
    x = torch.randn(224, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(5)).to(DEV)
    with torch.no_grad():
        m(x)                                       # (the first forward of a new batch size also runs mode auto's self-check probe: keep it out of the kernel table)
    with torch.no_grad(), ops.KernelTimer() as kt:
        fused = m(x)
    torch.cuda.synchronize()
    ks = kt.summary()
    assert "pv_token_norm" not in ks and ks["pv_rank_topk"]["launches"] == 3
    keep_f = [m.encoder.layers[li].last_keep.clone() for li in (3, 6, 9)]
    monkeypatch.setattr(engine, "_FUSE_RANK_NORM", False)
    with torch.no_grad(), ops.KernelTimer() as kt:
        plain = m(x)
    torch.cuda.synchronize()
    assert kt.summary()["pv_token_norm"]["launches"] == 3
    keep_p = [m.encoder.layers[li].last_keep for li in (3, 6, 9)]
    same = [torch.equal(a, b) for a, b in zip(keep_f, keep_p)]
    # the two norm computations sum the squares in different orders: only exact near-ties (1 ulp) may rank differently
    for a, b in zip(keep_f, keep_p):
        assert (a != b).float().mean().item() < 2e-3
    if all(same):
        assert torch.equal(fused, plain)


@pytest.mark.parametrize("M,N,K", [(2304, 768, 3072), (4000, 384, 1536)])
def test_gemm_rowsq_out(M, N, K):
    """pv_gemm_args.rowsq_out: per-column-tile sums of squares of the finished fp32 rows, vs the rows themselves."""
    from peekvit_amd import ops
    from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
    bias, res = torch.randn(N, generator=g, device=DEV), torch.randn(M, N, generator=g, device=DEV)
    out, out2 = torch.empty((M, N), device=DEV), torch.empty((M, N), device=DEV)
    tiles = (N + 255) // 256
    rowsq = torch.full((tiles, M), float("nan"), device=DEV)
    ops.gemm(a, w, bias, out, PV_EPI_BIAS_RES_F32, res=res, rowsq_out=rowsq)          # the feature selects the 256-row tile kernel
    ops.gemm(a, w, bias, out2, PV_EPI_BIAS_RES_F32, res=res)
    assert torch.equal(out, out2)                                        # the extra reduction does not touch the output
    for t in range(tiles):
        ref = (out[:, t * 256:(t + 1) * 256].double() ** 2).sum(1)
        assert rel_l2(rowsq[t].double(), ref) < 1e-6
    keep = ops.rank_topk_partials(rowsq, M // 8, 8, 4)                   # 8-token "images": rank rows 1..7 of each by norm
    norms = out.double().pow(2).sum(1).sqrt().view(M // 8, 8)[:, 1:]
    assert torch.equal(keep.long().cpu(), norms.argsort(dim=1, descending=True, stable=True)[:, :4].cpu())


def test_graph_owns_its_scratch():
    """A captured forward must survive later eager forwards that grow the shared workspace arena (the graph's nodes point into a
    private arena that the GraphedForward object keeps alive)."""
    from peekvit_amd.graph import GraphedForward
    cfg, m = _model("vit", "vit_tiny")
    x = _x(cfg, 4).to(DEV)
    with torch.no_grad():
        eager = m(x).clone()
    g = GraphedForward(m, x)
    big = torch.randn(24, 3, cfg["image_size"], cfg["image_size"], device=DEV)
    with torch.no_grad():
        for _ in range(2):
            m(big)                                   # replaces every shared scratch buffer by a larger one
        junk = [torch.randn(1 << 20, device=DEV) for _ in range(8)]      # and lets the allocator hand the freed blocks out again
    assert torch.equal(g(x), eager)
    del junk


def test_two_stream_forward_is_bit_identical(monkeypatch):
    """PEEKVIT_AMD_STREAMS=2 (opt-in): two half-batches on two HIP streams give the same bits."""
    from peekvit_amd import engine
    cfg, m = _model("vit", "vit_tiny")
    x = torch.randn(300, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(9)).to(DEV)
    with torch.no_grad():
        one = m(x).clone()
        monkeypatch.setattr(engine, "_STREAMS", 2)
        two = m(x)
        torch.cuda.synchronize()
        assert torch.equal(one, two)
        with torch.no_grad():
            m.encoder.layers[0].mlp.fc1.weight.mul_(1.0)          # new parameter version: the cast happens on one stream, both use it
        assert torch.equal(m(x), one)


def test_device_prefetcher_feeds_the_same_batches():
    """harness.pipeline.DevicePrefetcher: host batches copied one ahead on a side stream through a reused ring - every batch arrives intact
    and in order (fp32 NCHW and uint8 NHWC, pinned and pageable, a ragged last batch), and the evaluation loop built on it reports the
    reference loop's accuracy."""
    from peekvit_amd.harness.pipeline import DevicePrefetcher
    from peekvit_amd.harness import test as htest
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(5 if i < 6 else 3, 3, 32, 32, generator=g), torch.randint(0, 10, (5 if i < 6 else 3,), generator=g)) for i in range(7)]
    for depth in (1, 2):
        for pin in (False, True):
            src = [(x.pin_memory(), y.pin_memory()) if pin else (x, y) for x, y in batches]
            got = []
            for x, y in DevicePrefetcher(src, DEV, depth=depth):
                assert x.is_cuda and y.is_cuda
                got.append((x.square().sum(), y.sum()))            # work on the compute stream that reads the slot
            assert len(got) == len(batches)
            for (sx, sy), (x, y) in zip(got, batches):
                assert abs(float(sx) - float(x.double().square().sum())) < 1e-2 and int(sy) == int(y.sum())
    u8 = [(torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8), torch.zeros(4, dtype=torch.int64)) for _ in range(3)]
    for (x, _), (xr, _) in zip(DevicePrefetcher(u8, DEV), u8):
        assert torch.equal(x.cpu(), xr)
    cfg, m = _model("vit", "vit_micro")
    data = [(torch.from_numpy(synth.synth_images(4, cfg["image_size"], seed=s)), torch.randint(0, cfg["num_classes"], (4,), generator=g)) for s in range(5)]
    a = htest.evaluate(m, data, torch.device(DEV), [None], 20, prefetch=True)[0]
    b = htest.evaluate(m, data, torch.device(DEV), [None], 20, prefetch=False)[0]
    assert a["accuracy"] == b["accuracy"] and a["device_images_per_second"] > 0


def test_inference_mode_and_no_grad_agree_and_mix():
    """torch.inference_mode() (tensors without a version counter, buffers that must not become inference tensors) gives the logits of
    torch.no_grad() bit for bit - at a batch that takes the folded-LayerNorm hand-offs, first use of the model under inference_mode -
    the same model then still runs under no_grad and trains, and a hook that edits a block's output in place is honoured in both modes."""
    from peekvit_amd import engine
    engine.workspace.clear()
    cfg, m = _model("vit", "vit_b_16")
    x = _x(cfg, 56).to(DEV)
    with torch.inference_mode():
        a = m(x).clone()
    with torch.no_grad():
        b = m(x)
    assert torch.equal(a, b)
    h = m.encoder.layers[3].register_forward_hook(lambda mod, inp, out: out.mul_(0.5))     # in-place edit between two blocks
    try:
        with torch.inference_mode():
            c = m(x).clone()
        with torch.no_grad():
            d = m(x)
    finally:
        h.remove()
    assert torch.equal(c, d) and not torch.equal(c, a)
    rcfg, r = _model("rank", "vit_b_16", rankvit_layers=[3, 6, 9])
    r.set_budget(0.5)
    with torch.inference_mode():
        e = r(x).clone()
    with torch.no_grad():
        f = r(x)
    assert torch.equal(e, f)
    m.train()
    loss = torch.nn.functional.cross_entropy(m(x[:4]), torch.arange(4, device=DEV))
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_half_precision_model_is_served_from_fp32_copies():
    """model.half() / model.bfloat16() (16-bit PARAMETERS and inputs): the kernels read biases / LayerNorm affine / tokens as fp32 and make
    their operand copies of the weights from the stored values - the logits are those of the fp32 model holding the same (rounded) values."""
    cfg, m = _model("vit", "vit_tiny")
    x = _x(cfg, 3).to(DEV)
    for conv in (lambda t: t.half(), lambda t: t.bfloat16()):
        m16 = conv(_model("vit", "vit_tiny")[1])
        ref = _model("vit", "vit_tiny")[1]
        ref.load_state_dict({k: v.float() for k, v in m16.state_dict().items()})
        with torch.no_grad():
            got = m16(conv(x))
            want = ref(conv(x).float())
        assert got.dtype == torch.float32 and torch.equal(got, want)


def test_empty_batch_returns_empty_logits():
    """A batch of zero images (the tail of a sharded loader): [0, num_classes] like the reference, nothing is launched."""
    from peekvit_amd import ops
    cfg, m = _model("vit", "vit_tiny")
    n0 = ops.launch_count
    with torch.no_grad():
        y = m(torch.empty(0, 3, cfg["image_size"], cfg["image_size"], device=DEV))
    assert y.shape == (0, cfg["num_classes"]) and ops.launch_count == n0


def test_two_threads_forward_concurrently():
    """Two Python threads run guarded forwards at the same time on the same GPU and (default) stream - one of them keeps tripping the fp16 range
    guard and repeating on bf16 operands, i.e. switching the operand library around its forwards.  Mode, operand library, range flag and
    scratch arena are per thread: each thread's logits are the ones it gets alone."""
    import threading
    import warnings
    cfg, m_ok = _model("vit", "vit_small")
    _, m_ovf = _model("vit", "vit_small")
    with torch.no_grad():
        m_ovf.encoder.layers[1].mlp.fc1.bias.add_(1e5)           # GELU outputs beyond fp16: every forward of this model falls back to bf16
    xa, xb = _x(cfg, 24).to(DEV), _x(cfg, 17).to(DEV)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_a, ref_b = m_ok(xa).clone(), m_ovf(xb).clone()
    out, err = {}, []

    def work(name, model, x, ref):
        try:
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(12):
                    y = model(x)
                    if not torch.equal(y, ref):
                        err.append(name)
                        return
            out[name] = True
        except Exception as e:                                    # noqa: BLE001 - reported below
            err.append(f"{name}: {type(e).__name__}: {e}")

    ts = [threading.Thread(target=work, args=("fp16 thread", m_ok, xa, ref_a)), threading.Thread(target=work, args=("fallback thread", m_ovf, xb, ref_b))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err and len(out) == 2, err


def test_rankvit_budget_zero_keeps_the_class_token_only():
    """models/rankvit.py:74 keeps ceil(N * budget) tokens: budget 0 drops every patch token, the class token alone goes on."""
    import os
    cfg, m = _model("rank", "vit_tiny", rankvit_layers=[1, 2])
    m.set_budget(0.0)
    x = _x(cfg, 3).to(DEV)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (3, cfg["num_classes"]) and torch.isfinite(y).all() and m.encoder.layers[1].last_keep.shape == (3, 0)
    os.environ["PEEKVIT_AMD_BACKEND"] = "torch"
    try:
        with torch.no_grad():
            ref = m(x)
    finally:
        del os.environ["PEEKVIT_AMD_BACKEND"]
    assert rel_l2(y.cpu(), ref.cpu()) < 5e-3
    # ... and against the CPU ORACLE (rankvit.py:74 with budget 0 in the reference's arithmetic) at the contract's tolerance (round 3)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    assert rel_l2(y.cpu(), O.vit_forward(x.cpu(), sd, cfg, "fp32", rankvit_layers=[1, 2], budget=0.0)) < TOL_CONTRACT
    m.train()
    torch.nn.functional.cross_entropy(m(x), torch.arange(3, device=DEV)).backward()
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())


@pytest.mark.parametrize("kind,extra,budget", [("rank", {"rankvit_layers": [1, 3]}, 0.5),
                                               ("res", dict(gate_type="sigmoid", gate_temp=1, gate_bias=10, add_budget_token="learnable", gate_threshold=0.5), 0.5)])
def test_hip_graph_replay_of_the_pruning_models(kind, extra, budget):
    """RankViT (ranking + compaction) and ResidualViT (gate + masked blocks) forwards are capturable as one hipGraph: no host read of a device
    value inside them (the budget is kept as a host float by set_budget); replay is bit-identical to eager, also after new input."""
    from peekvit_amd.graph import GraphedForward
    cfg, m = _model(kind, "vit_tiny", **extra)
    m.set_budget(budget)
    x0, x1 = _x(cfg, 6).to(DEV), torch.from_numpy(synth.synth_images(6, cfg["image_size"], seed=9)).to(DEV)
    with torch.no_grad():
        g = GraphedForward(m, x0)
        for x in (x0, x1, x0):
            assert torch.equal(g(x).clone(), m(x))


def test_patch_size_whose_columns_are_not_a_multiple_of_64():
    """P = 14 (ViT-H/14-style stems: 3 * 14 * 14 = 588 columns per patch, 257 tokens at 224 x 224): the patch-embedding GEMM runs on zero-padded
    columns; logits against the stock-op composite of the same module."""
    import os
    from peekvit_amd.models.vit import VisionTransformer
    torch.manual_seed(0)
    m = VisionTransformer(image_size=224, patch_size=14, num_layers=2, num_heads=4, hidden_dim=256, mlp_dim=1024, num_classes=50).to(DEV).eval()
    with torch.no_grad():
        for p in m.parameters():
            if float(p.abs().sum()) == 0.0:
                p.normal_(std=0.02)
        x = torch.randn(5, 3, 224, 224, device=DEV)
        y = m(x)
        os.environ["PEEKVIT_AMD_BACKEND"] = "torch"
        try:
            ref = m(x)
        finally:
            del os.environ["PEEKVIT_AMD_BACKEND"]
    assert rel_l2(y.cpu(), ref.cpu()) < 2e-3
    # ... and against the CPU ORACLE at the contract's tolerance (round 3)
    cfg = dict(image_size=224, patch_size=14, num_layers=2, num_heads=4, hidden_dim=256, mlp_dim=1024, num_classes=50)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    assert rel_l2(y.cpu(), O.vit_forward(x.cpu(), sd, cfg, "fp32")) < TOL_CONTRACT


# ---- round 3: the matrix of arithmetic-variant switches, each pinned to the oracle ---------------------------------------------------------
# Every module-level switch of peekvit_amd.engine that selects a DIFFERENT sequence of kernels for the same logits (and the persistent GEMM
# launch of the library) against the CPU oracle at the contract's tolerance, on two widths: 384 (full-row GEMM with the fused LayerNorm) and
# 768 (LayerNorm folded into the GEMM epilogues), at a batch where the 256-row kernels, the folds and the class-row last block all engage.
_VARIANT_CFGS = {
    "w384": dict(image_size=224, patch_size=16, num_layers=2, num_heads=6, hidden_dim=384, mlp_dim=1536, num_classes=100),
    "w768": dict(image_size=224, patch_size=16, num_layers=2, num_heads=12, hidden_dim=768, mlp_dim=3072, num_classes=100),
}
_VARIANTS = [("default", None, None), ("no LayerNorm folding", "_FOLD_LN", False), ("no full-row GEMM", "_FULLROW_LN", False),
             ("row-block fused LayerNorm", "_FUSE_LN", True), ("no split-K", "_SMALL_M_SPLITK", False), ("last block on all rows", "_LAST_BLOCK_ROWS", False),
             ("two streams", "_STREAMS", 2), ("large batch: persistent GEMM launch", "BIG", None), ("large batch, one tile per workgroup", "PF", 0)]
_variant_ref = {}


@pytest.mark.parametrize("width", list(_VARIANT_CFGS))
@pytest.mark.parametrize("label,flag,value", _VARIANTS)
def test_every_arithmetic_variant_switch_meets_the_contract(monkeypatch, width, label, flag, value):
    from peekvit_amd import _lib, engine
    from peekvit_amd.models.vit import VisionTransformer
    cfg = _VARIANT_CFGS[width]
    # the stream split needs >= 256 images (engine._STREAMS_MIN_BATCH); the persistent GEMM launch needs >= 512 tiles (so does its "off" switch)
    B = 16 if flag not in ("_STREAMS", "PF", "BIG") else 512 if width == "w384" else 256
    m = VisionTransformer(**cfg)
    synth.load_synth_weights(m, cfg, "vit", seed=0)
    m = m.eval().to(DEV)
    x = torch.from_numpy(synth.synth_images(B, cfg["image_size"], seed=3)).to(DEV)
    key = (width, B)
    if key not in _variant_ref:                     # the oracle on the first 16 images (a forward is per-image: any batch's first rows must agree)
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        _variant_ref[key] = O.vit_forward(x[:16].cpu(), sd, cfg, "fp32")
    lib16 = _lib.load("f16")
    if flag == "PF":
        lib16.pv_debug_set_gemm_pf(value)
    elif flag is not None and flag != "BIG":
        monkeypatch.setattr(engine, flag, value)
    try:
        n0 = ops_launches()
        with torch.no_grad():
            y = m(x)
        assert ops_launches() > n0
    finally:
        if flag == "PF":
            lib16.pv_debug_set_gemm_pf(-1)
    err = rel_l2(y[:16].cpu(), _variant_ref[key])
    assert err < TOL_CONTRACT, (width, label, err)


def ops_launches():
    from peekvit_amd import ops
    return ops.launch_count


@pytest.mark.parametrize("kind", ["res", "rank", "vit_last"])
def test_local_fallback_of_the_score_guard_on_other_model_families(kind, monkeypatch):
    """Round 5: the LOCAL fallback (hybrid layers) on ResidualViT's gated blocks (the gate kernel's LayerNorm hand-off and the
    never-materialised masked tokens have to step aside for a hybrid layer), on RankViT (shorter sequences behind a ranked layer) and on a
    ViT whose LAST block - otherwise computed for the class rows only - is the one with the large scores.  One layer's q and k rows are scaled
    so that its attention logits reach ~50: the score guard names exactly that layer, the forward is repeated once with its attention half in
    split precision, and the logits meet the contract against the fp32 CPU oracle."""
    import warnings
    from peekvit_amd import engine
    name = "vit_tiny"
    layer = 3 if kind == "vit_last" else 1
    extra = dict(gate_type="sigmoid", gate_temp=1, gate_bias=2, add_budget_token="learnable", gate_threshold=0.5) if kind == "res" else \
        (dict(rankvit_layers=[2]) if kind == "rank" else {})
    cfg, m = _model("vit" if kind == "vit_last" else kind, name, **extra)
    D = cfg["hidden_dim"]
    with torch.no_grad():
        mha = m.encoder.layers[layer].self_attention.self_attention
        gain = 12.0 if kind == "res" else 5.0                     # (ResidualViT: the gate's mask ~0.45 scales the LayerNorm output, i.e. q and k, down)
        mha.in_proj_weight[:2 * D] *= gain
        mha.in_proj_bias[:2 * D] *= gain
    engine.reset_guard(m)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    x = _x(cfg, 4)
    if kind in ("res", "rank"):
        m.set_budget(0.5)
    h0, f0 = engine.hybrid_fallback_count, engine.fallback_count
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        logits = m(x.to(DEV)).cpu().numpy()
        again = m(x.to(DEV)).cpu().numpy()                       # the layers are remembered: no second trip, same bits
    st = engine.guard_state(m)
    assert sorted(st.hybrid) == [layer], sorted(st.hybrid)
    assert engine.hybrid_fallback_count == h0 + 1 and np.array_equal(again, logits)
    # On a model this small a sharp layer amplifies the 16-bit noise of everything in FRONT of it (a score of 50 turns 1e-4 of input noise into
    # half a percent of a probability) - no arithmetic inside the layer can take that back.  The self-check measures it: where the hybrid forward
    # is outside 9e-4 it answers from the split-operand arithmetic for the whole forward (all three vit_tiny cases here: 1.6e-3, 9.2e-4, 1.8e-3
    # measured), where it is inside (the trained-like ViT-B/16 fixture: 2.0e-4, tests/test_hip_precision.py) the hybrid forward answers.
    answered_by_x3 = engine.fallback_count > f0
    print(kind, "hybrid layer", layer, "answered by", "bf16x3" if answered_by_x3 else "the hybrid forward", "self-check", engine.selfcheck_last)
    if kind == "res":
        ref = O.residualvit_forward(x, sd, dict(cfg, **extra), 0.5, "fp32").numpy()
    elif kind == "rank":
        ref = O.vit_forward(x, sd, cfg, "fp32", rankvit_layers=[2], budget=0.5).numpy()
    else:
        ref = O.vit_forward(x, sd, cfg, "fp32").numpy()
    assert rel_l2(logits, ref) < TOL_CONTRACT, rel_l2(logits, ref)
    engine.reset_guard(m)


def test_rank_topk_reports_the_gap_at_the_keep_boundary():
    """pv_rank_topk_gap / pv_rank_topk_partials_gap (ABI v10): gap_min[b] is lowered to the relative gap between image b's last kept and first dropped
    norm; the kept indices are those of the plain entry points; an array handed to two rankings keeps the smaller gap; k == N leaves it alone."""
    from peekvit_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N, k = 37, 196, 98
    norms = (torch.rand(B, N, generator=g) + 0.5).to(DEV)
    gap = torch.full((B,), float("inf"), device=DEV)
    keep = ops.rank_topk(norms, k, gap)
    assert torch.equal(keep, ops.rank_topk(norms, k))
    srt = torch.sort(norms.cpu().double(), dim=1, descending=True).values
    want = (srt[:, k - 1] - srt[:, k]) / srt[:, k - 1]
    assert torch.allclose(gap.cpu().double(), want, rtol=1e-5, atol=1e-7)
    k2 = 150
    ops.rank_topk(norms, k2, gap)                                   # a second ranking into the same array: the minimum of both
    want2 = torch.minimum(want, (srt[:, k2 - 1] - srt[:, k2]) / srt[:, k2 - 1])
    assert torch.allclose(gap.cpu().double(), want2, rtol=1e-5, atol=1e-7)
    before = gap.clone()
    ops.rank_topk(norms, N, gap)                                    # nothing dropped: no boundary
    assert torch.equal(gap, before)
    # the partials form: norms = sqrt of per-tile sums of squares over rows [B * S], class row first
    S, tiles = N + 1, 3
    rowsq = torch.rand(tiles, B * S, generator=g).to(DEV)
    gp = torch.full((B,), float("inf"), device=DEV)
    kp = ops.rank_topk_partials(rowsq, B, S, k, gp)
    assert torch.equal(kp, ops.rank_topk_partials(rowsq, B, S, k))
    nn_ = rowsq.cpu().double().sum(0).sqrt().view(B, S)[:, 1:]
    s2 = torch.sort(nn_, dim=1, descending=True).values
    assert torch.allclose(gp.cpu().double(), (s2[:, k - 1] - s2[:, k]) / s2[:, k - 1], rtol=1e-4, atol=1e-6)


def test_rankvit_repairs_the_images_that_sit_on_a_ranking_near_tie(monkeypatch):
    """Round 6 (review item 6): with engine.RANK_REPAIR the forward watches every image's relative gap at its keep boundaries and re-runs the images under
    engine.RANK_TIE_GAP - where 16-bit noise in the norms may have kept another token than the reference's fp32 ranking - in the split-operand arithmetic:
    their kept sets and logits become that arithmetic's, every other image stays on fp16 operands.  Threshold 0: nothing changes."""
    from peekvit_amd import engine
    cfg, m = _model("rank", "vit_b_16", rankvit_layers=[3, 6, 9])
    m.set_budget(0.5)
    B = 48
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(21)).to(torch.bfloat16).float().to(DEV)
    ranked = [m.encoder.layers[i] for i in (3, 6, 9)]
    sets = lambda: [torch.sort(b.last_keep, dim=1).values.clone() for b in ranked]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with engine.precision("f16"), engine.rank_gaps(B, x.device) as gap:
            plain = m(x).clone()
            plain_sets = sets()
        with engine.precision("bf16x3"):
            exact = m(x).clone()
            exact_sets = sets()
        flipped = torch.zeros(B, dtype=torch.bool, device=DEV)
        for a, b in zip(plain_sets, exact_sets):
            flipped |= (a != b).any(dim=1)
        thr = float(torch.sort(gap).values[B // 3])                 # a third of the images count as near-ties
        monkeypatch.setattr(engine, "RANK_REPAIR", True)
        monkeypatch.setattr(engine, "RANK_TIE_GAP", thr)
        monkeypatch.setattr(engine, "SELFCHECK_IMAGES", 0)          # (the arithmetic under test is the repair, not the self-check's verdict)
        engine.reset_guard(m)
        r0 = engine.rank_repaired_images
        got = m(x)
        got_sets = sets()
        near = gap < thr
        n_near = int(near.sum())
        assert 0 < n_near < B // 2 and engine.rank_repaired_images == r0 + n_near
        for a, b, c in zip(got_sets, exact_sets, plain_sets):
            assert torch.equal(a[near], b[near]) and torch.equal(a[~near], c[~near])      # repaired images: the exact sets; the others: untouched
        assert torch.equal(got[~near], plain[~near])
        assert rel_l2(got[near], exact[near]) < 2e-5                                       # (a sub-batch of another size: fp32 summation order of the split GEMMs)
        # a threshold of 0: nothing is repaired
        monkeypatch.setattr(engine, "RANK_TIE_GAP", 0.0)
        engine.reset_guard(m)
        assert torch.equal(m(x), plain) and engine.rank_repaired_images == r0 + n_near
        # a dense boundary (more than half of the batch under the threshold): the whole batch runs in split precision
        monkeypatch.setattr(engine, "RANK_TIE_GAP", float(torch.sort(gap).values[-2]))
        engine.reset_guard(m)
        dense = m(x)
        assert rel_l2(dense, exact) < 2e-5 and engine.rank_repaired_images == r0 + n_near + B
        for a, b in zip(sets(), exact_sets):
            assert torch.equal(a, b)
    assert bool(flipped.any()) or True      # (random weights on random images: some image usually flips; the assertions above do not depend on it)


def test_rank_strict_holds_on_random_images_through_the_near_tie_repair(monkeypatch):
    """PEEKVIT_AMD_RANK_STRICT=1 (a ranking tie flip counts as a contract violation) implies the repair: on a batch of random images - where the
    synthetic model's keep boundary is dense - the images under the calibrated gap run in split precision, the self-check's probe images keep the
    split-operand arithmetic's token sets (no flip left to count), the model is NOT sent to bf16x3 for good, and over the whole batch the kept sets are
    the exact ones (profiles/r06_rank_tie_calibration.json: the largest gap of a flipped image among 1 024 was 9.6e-4, the threshold is 1.2e-3)."""
    from peekvit_amd import engine
    monkeypatch.setattr(engine, "RANK_STRICT", True)
    monkeypatch.setattr(engine, "RANK_REPAIR", True)
    cfg, m = _model("rank", "vit_b_16", rankvit_layers=[3, 6, 9])
    m.set_budget(0.5)
    B = 512
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(33)).to(torch.bfloat16).float().to(DEV)
    ranked = [m.encoder.layers[i] for i in (3, 6, 9)]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")                      # (a flip warning or a verdict warning would be an error here)
        t0, c0 = engine.selfcheck_trips, engine.selfcheck_count
        got = m(x)
        got_sets = [torch.sort(b.last_keep, dim=1).values.clone() for b in ranked]
        assert engine.selfcheck_count == c0 + 1 and engine.selfcheck_trips == t0 and engine.selfcheck_last[2] == 0 and engine.selfcheck_last[0] < engine.SELFCHECK_LIMIT
        assert not engine.guard_state(m).unsafe
        with engine.precision("bf16x3"):
            exact = m(x)
            exact_sets = [torch.sort(b.last_keep, dim=1).values.clone() for b in ranked]
    wrong = torch.zeros(B, dtype=torch.bool, device=DEV)
    for a, b in zip(got_sets, exact_sets):
        wrong |= (a != b).any(dim=1)
    assert int(wrong.sum()) <= 1                            # (an image beyond the calibrated gap that still flips: none among 1 024 in the calibration run)
    assert rel_l2(got[~wrong], exact[~wrong]) < TOL_CONTRACT
