"""Host-side logic added in round 6 that needs no GPU: the power / clock sampler bench.py brackets its timed regions with, and the rule by which mode
"auto" decides that a model forward is launch-bound (peekvit_amd/autograph.py)."""
import time

import torch

from peekvit_amd import autograph, synth, telemetry
from peekvit_amd.models.vit import VisionTransformer


def test_power_sampler_reports_nothing_rather_than_failing_without_hwmon(tmp_path, monkeypatch):
    """A host without the sysfs files (this container): the sampler thread runs, a window's result says `samples: 0`-style emptiness or whatever cards it
    found, and nothing raises - bench.py must never lose its result line to telemetry."""
    s = telemetry.PowerSampler("0000:ff:1f.7", period_s=0.005)          # (no such device: falls back to every card, possibly none)
    s.start()
    with s.window() as w:
        time.sleep(0.05)
    s.stop()
    r = w.result()
    assert isinstance(r, dict) and "samples" in r and r["samples"] >= 0
    if not s.cards:
        assert r == {"samples": r["samples"]} and s.cap_w() is None
    else:
        assert r["samples"] > 0 and ("power_w" in r or "sclk_mhz" in r)


def test_power_sampler_reads_a_card_it_is_pointed_at(tmp_path, monkeypatch):
    """The sampler against a fake hwmon tree: microwatts -> watts, hertz -> MHz, the window's mean and extremes, the cap."""
    card = tmp_path / "card0" / "device" / "hwmon" / "hwmon3"
    card.mkdir(parents=True)
    (card / "power1_average").write_text("1250000000\n")
    (card / "freq1_input").write_text("2100000000\n")
    (card / "power1_cap").write_text("1400000000\n")
    monkeypatch.setattr(telemetry.glob, "glob", lambda pat: [str(card)] if "hwmon" in pat else [])
    s = telemetry.PowerSampler(None, period_s=0.002)
    assert len(s.cards) == 1
    s.start()
    with s.window() as w:
        time.sleep(0.03)
        (card / "power1_average").write_text("1350000000\n")
        time.sleep(0.03)
    s.stop()
    r = w.result()
    assert r["samples"] >= 4 and 1250.0 <= r["power_w"] <= 1350.0 and r["power_w_max"] == 1350.0 and r["sclk_mhz"] == 2100.0 and s.cap_w() == 1400.0
    assert w.result(skip_frac=0.9)["power_w"] == 1350.0


def test_launch_bound_rule_picks_the_small_shapes():
    """autograph.launch_bound: GEMM + attention FLOPs of the encoder against the host time of an eager forward - vit_tiny at batch 32 (BASELINE config 1's
    model: 0.59 ms by launches, 0.41 ms as one replay) and single-digit batches of the larger models; never the benchmark configurations."""
    def lb(name, b):
        return autograph.launch_bound(VisionTransformer(**synth.MODEL_CONFIGS[name]), b)
    assert lb("vit_tiny", 32) and not lb("vit_tiny", 128)
    assert lb("vit_small", 8) and not lb("vit_small", 512)
    assert lb("vit_b_16", 1) and lb("vit_b_16", 8) and not lb("vit_b_16", 64) and not lb("vit_b_16", 2048)
    assert not autograph.launch_bound(torch.nn.Linear(4, 4), 1)          # (not one of the model classes: no attributes to estimate from)


def test_new_entry_points_validate_their_arguments_before_touching_the_gpu():
    """pv_patch_embed_f32 / pv_rank_topk_gap / pv_rank_topk_partials_gap (ABI v10) refuse null pointers, misaligned buffers and shapes outside their contract
    with an error code - no launch, so this runs without a GPU."""
    import ctypes as C
    from peekvit_amd import _lib
    lib = _lib.load()
    INVALID, UNSUPPORTED = -1, -2
    z = C.c_void_p(0)
    a = C.c_void_p(4096)           # (never dereferenced: every call below is refused first)
    assert lib.pv_patch_embed_f32(z, a, a, a, a, 2, 3, 224, 16, 384, 197, 1, z, z) == INVALID            # no image
    assert lib.pv_patch_embed_f32(a, a, a, a, a, 2, 3, 224, 16, 384, 150, 1, z, z) == INVALID            # 1 + 196 patch rows do not fit S = 150
    assert lib.pv_patch_embed_f32(a, a, a, a, a, 2, 3, 224, 14, 384, 257, 1, z, z) == UNSUPPORTED        # P = 14: not a multiple of 8
    assert lib.pv_patch_embed_f32(a, a, a, a, a, 2, 3, 225, 16, 384, 197, 1, z, z) == UNSUPPORTED        # R % P
    assert lib.pv_patch_embed_f32(C.c_void_p(4100), a, a, a, a, 2, 3, 224, 16, 384, 197, 1, z, z) == UNSUPPORTED      # misaligned image
    assert lib.pv_rank_topk_gap(z, a, z, 4, 196, 98, z) == INVALID and lib.pv_rank_topk_gap(a, a, z, 4, 196, 197, z) == INVALID      # k > N
    assert lib.pv_rank_topk_gap(a, a, a, 4, 5000, 98, z) == UNSUPPORTED                                   # N > 4096
    assert lib.pv_rank_topk_partials_gap(a, 3, a, a, 4, 1, 0, z) == INVALID                               # S < 2
    assert lib.pv_rank_topk_partials_gap(a, 100, a, a, 4, 197, 98, z) == UNSUPPORTED                      # more than 64 column tiles
    assert lib.pv_version() == 10
