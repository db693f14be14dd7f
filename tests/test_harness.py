"""Harness counterparts on CPU: config loader semantics, eval loop, two training steps, checkpoint round trip."""
import json
import os

import torch

from peekvit_amd.harness import checkpoint, config, test as htest, train as htrain

MICRO = ["model=vit_tiny", "model.patch_size=8", "model.hidden_dim=64", "model.mlp_dim=128", "model.num_layers=2", "model.num_heads=2",
         "dataset.image_size=32", "dataset.num_classes=10", "dataset.train_size=16", "dataset.val_size=16", "device=cpu"]


def test_config_defaults_overrides_interpolation():
    cfg = config.load_config("test_config", ["model=vit_b_16", "test.test_batch_size=2048", "dataset.num_classes=10"])
    assert cfg["model"]["_target_"] == "peekvit.models.vit.VisionTransformer" and cfg["model"]["hidden_dim"] == 768
    assert cfg["model"]["num_classes"] == 10 and cfg["model"]["image_size"] == 224          # ${dataset.*} resolved after overrides
    assert cfg["test"]["test_batch_size"] == 2048 and cfg["device"] == "cuda:0"
    m = config.instantiate(config.load_config("test_config", MICRO)["model"])
    assert type(m).__name__ == "VisionTransformer" and m.hidden_dim == 64


def test_eval_loop_reports_reference_metrics():
    res = htest.main(MICRO + ["test.test_batch_size=8"])
    assert len(res) == 1 and res[0]["images_per_second"] > 0 and 0.0 <= res[0]["accuracy"] <= 1.0 and res[0]["flops_per_image"] > 0
    res = htest.main(MICRO + ["model=rankvit_b_16", "model.rankvit_layers=[1]", "test.budgets=[0.5,1.0]"])
    assert [r["budget"] for r in res] == [0.5, 1.0] and res[0]["flops_per_image"] < res[1]["flops_per_image"]


def test_train_steps_and_checkpoint_round_trip(tmp_path):
    hist = htrain.main(MICRO + ["training.train_batch_size=8", "training.num_epochs=2", f"experiment_dir={tmp_path}"])
    assert len(hist["loss"]) == 2 and all(l == l for l in hist["loss"])            # finite
    path = checkpoint.get_checkpoint_path(str(tmp_path))
    assert path.endswith("epoch_001.pth")
    state = torch.load(path, weights_only=False)
    assert set(state) == {"model_class", "noise_args", "model_args", "state_dict", "optimizer", "epoch"}
    model, _ = checkpoint.load_state(path)                                          # rebuilt from model_class / model_args
    assert type(model).__name__ == "VisionTransformer" and model.hidden_dim == 64
    res = htest.main(MICRO + [f"load_from={tmp_path}"])
    assert res[0]["accuracy"] == hist["val_accuracy"][-1]


def test_loads_a_checkpoint_shaped_like_the_reference_writes_it(tmp_path):
    """utils/utils.py:198-215 stores model_args = dict(cfg.model): Hydra's `_target_` and the pretrained-weight keys included,
    optimizer None (skip_optimizer=True); utils/utils.py:236-238 pops those keys before rebuilding and loads with strict=False."""
    from peekvit_amd.models.vit import VisionTransformer
    args = dict(image_size=32, patch_size=8, num_layers=2, num_heads=2, hidden_dim=64, mlp_dim=128, num_classes=10)
    src = VisionTransformer(**args)
    torch.nn.init.normal_(src.head.weight, std=0.02)
    ref_style = {"model_class": "VisionTransformer", "noise_args": None,
                 "model_args": dict(args, _target_="peekvit.models.vit.VisionTransformer", torch_pretrained_weights="ViT_B_16_Weights.IMAGENET1K_V1",
                                    timm_pretrained_weights=None, dropout=0.0, attention_dropout=0.0),
                 "state_dict": src.state_dict(), "optimizer": None, "epoch": 7}
    os.makedirs(tmp_path / "checkpoints")
    torch.save(ref_style, tmp_path / "checkpoints" / "epoch_007.pth")
    model, state = checkpoint.load_state(checkpoint.get_checkpoint_path(str(tmp_path)))
    assert state["epoch"] == 7 and all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), src.state_dict().values()))
    # strict=False like the reference: a checkpoint of a plain ViT loads into a model with extra parameters
    sd = dict(src.state_dict()); sd.pop("head.bias")
    torch.save(dict(ref_style, state_dict=sd), tmp_path / "checkpoints" / "epoch_008.pth")
    model2, _ = checkpoint.load_state(checkpoint.get_checkpoint_path(str(tmp_path)))
    assert torch.equal(model2.head.weight, src.head.weight)


def test_training_resumes_with_optimizer_state(tmp_path):
    base = MICRO + ["training.train_batch_size=8", f"experiment_dir={tmp_path}"]
    htrain.main(base + ["training.num_epochs=1"])
    st = torch.load(checkpoint.get_checkpoint_path(str(tmp_path)), weights_only=False)
    assert st["optimizer"] is not None and st["model_args"]["_target_"].endswith("VisionTransformer")       # stored as the reference stores it
    hist = htrain.main(base + ["training.num_epochs=3", f"resume_from={tmp_path}"])
    assert len(hist["loss"]) == 2 and checkpoint.get_checkpoint_path(str(tmp_path)).endswith("epoch_002.pth")   # epochs 1 and 2 only


def test_load_from_initialises_weights_only_like_the_reference(tmp_path):
    """train/train.py:64-70: `load_from` (a .pth file or an experiment directory) loads the WEIGHTS with strict=False, ignores optimizer state
    and epoch, and training still runs epochs 0..N - the reference's way to fine-tune a RankViT / ResidualViT from a finished ViT run."""
    import pytest
    src_dir, dst_dir = tmp_path / "vit", tmp_path / "rank"
    htrain.main(MICRO + ["training.train_batch_size=8", "training.num_epochs=2", f"experiment_dir={src_dir}"])
    ck = checkpoint.get_checkpoint_path(str(src_dir))
    assert ck.endswith("epoch_001.pth")
    # (1) a directory: another model family (extra / different parameters: the ViT's optimizer state could not even be loaded), all epochs run
    hist = htrain.main(MICRO + ["model=rankvit_b_16", "model.rankvit_layers=[1]", "training.train_batch_size=8", "training.num_epochs=2",
                                f"experiment_dir={dst_dir}", f"load_from={src_dir}"])
    assert len(hist["loss"]) == 2 and checkpoint.get_checkpoint_path(str(dst_dir)).endswith("epoch_001.pth")
    # (2) a .pth path is taken as the file itself; the weights really arrive (0 epochs: the saved model equals the checkpoint's)
    from peekvit_amd.harness.config import instantiate, load_config
    model = instantiate(load_config("train_config", MICRO)["model"])
    checkpoint.load_state(ck, model=model)
    want = {k: v.clone() for k, v in model.state_dict().items()}
    hist = htrain.main(MICRO + ["training.train_batch_size=8", "training.num_epochs=0", f"load_from={ck}"])
    assert hist["loss"] == []
    # (3) set but nothing there: an error, never a silent training run from scratch
    with pytest.raises(FileNotFoundError):
        htrain.main(MICRO + ["training.num_epochs=1", f"load_from={tmp_path / 'nowhere'}"])
    with pytest.raises(FileNotFoundError):
        htrain.main(MICRO + ["training.num_epochs=1", f"load_from={tmp_path / 'nowhere.pth'}"])
    assert all(torch.equal(want[k], v) for k, v in torch.load(ck, weights_only=False)["state_dict"].items())


def test_noise_block_matches_the_reference_semantics():
    """models/blocks.py:100-186 restated independently here: gaussian noise at an SNR in dB relative to each token's power (snr 0 adds
    nothing), token_drop zeroes int(prob * S) positions drawn with torch.randperm and shared by the batch, std is refused."""
    import pytest
    from peekvit_amd.models.blocks import NoiseBlock
    x = torch.randn(3, 10, 8)
    nb = NoiseBlock("gaussian")
    nb.set_value(20.0)
    torch.manual_seed(5)
    got = nb(x)
    torch.manual_seed(5)
    want = x + torch.randn_like(x) * torch.sqrt((x ** 2).mean(-1, keepdim=True) / 10 ** (20.0 / 10))
    assert torch.equal(got, want)
    nb.set_value(0)
    torch.manual_seed(9)
    assert torch.equal(nb(x), x)
    after = torch.get_rng_state()
    torch.manual_seed(9)
    torch.randn_like(x)                   # snr 0 adds nothing but still DRAWS (reference blocks.py:127): a seeded sweep stays comparable after it
    assert torch.equal(after, torch.get_rng_state())
    snr = 10 * torch.log10((x ** 2).mean() / ((got - x) ** 2).mean())
    assert abs(float(snr) - 20.0) < 1.5                                              # the realised SNR is the requested one
    td = NoiseBlock("token_drop", prob=0.3)
    torch.manual_seed(7)
    got = td(x)
    torch.manual_seed(7)
    idx = torch.randperm(10)[:3]
    want = x.clone(); want[:, idx] = 0
    assert torch.equal(got, want) and int((got.abs().sum(-1) == 0).sum()) == 3 * 3
    td.set_value(0)
    assert td(x) is x
    with pytest.raises(ValueError):
        NoiseBlock("gaussian", std=0.1)
    with pytest.raises(AssertionError):
        td.set_snr(3.0)                                                               # a token_drop block has no SNR


def test_add_noise_splices_the_encoder_and_the_sweep_reports_every_value():
    from peekvit_amd.harness.noise import add_noise
    from peekvit_amd.models.blocks import NoiseBlock
    model = config.instantiate(config.load_config("test_config", MICRO)["model"])
    blocks = list(model.encoder.layers)
    nm = add_noise(model, layer=1, noise_type="token_drop", prob=0.5)
    layers = list(model.encoder.layers)
    assert isinstance(nm, NoiseBlock) and layers[1] is nm and layers[0] is blocks[0] and layers[2] is blocks[1] and len(layers) == 3
    assert "noise" not in "".join(model.state_dict().keys())                         # no parameters, nothing new in the state dict
    x = torch.randn(2, 3, 32, 32)
    torch.nn.init.normal_(model.head.weight, std=0.02)                                # (the reference's zero head would hide everything)
    with torch.no_grad():
        model.eval()
        torch.manual_seed(1); a = model(x)
        nm.set_value(0.0); b = model(x)
    assert a.shape == b.shape and not torch.allclose(a, b)
    # named layers (the reference's OrderedDict case): inserted under the name 'noise'
    from collections import OrderedDict
    model2 = config.instantiate(config.load_config("test_config", MICRO)["model"])
    model2.encoder.layers = torch.nn.Sequential(OrderedDict((f"encoder_layer_{i}", m) for i, m in enumerate(model2.encoder.layers)))
    add_noise(model2, layer=2, noise_type="gaussian", snr=10.0)
    assert [n for n, _ in model2.encoder.layers.named_children()] == ["encoder_layer_0", "encoder_layer_1", "noise"]
    # the harness loop: budgets x noise values, the clean-channel row equals the run without a noise module
    clean = htest.main(MICRO + ["test.test_batch_size=8"])
    res = htest.main(MICRO + ["test.test_batch_size=8", "noise=gaussian", "noise.layer=1", "test.noises=[0.0,-10.0]"])
    assert [r["noise"] for r in res] == [0.0, -10.0] and res[0]["noise_type"] == "gaussian"
    assert res[0]["accuracy"] == clean[0]["accuracy"] and res[0]["flops_per_image"] == clean[0]["flops_per_image"]
    # (the splice shifts the indices of the blocks behind it, and RankVisionTransformer.set_budget indexes encoder.layers by position -
    # models/rankvit.py:287-288 - so, as in the reference, the ranked layers must sit in front of the noise module)
    res = htest.main(MICRO + ["model=rankvit_b_16", "model.rankvit_layers=[0]", "test.budgets=[0.5,1.0]", "noise=digital", "noise.layer=1", "test.noises=[0.0,0.5]"])
    assert [(r["budget"], r["noise"]) for r in res] == [(0.5, 0.0), (0.5, 0.5), (1.0, 0.0), (1.0, 0.5)]
    assert res[0]["flops_per_image"] < res[2]["flops_per_image"]
