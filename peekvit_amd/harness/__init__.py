"""Harness counterparts of the reference's entry scripts: evaluation (validate/test.py) and
training (train/train.py) loops driven by a small Hydra-compatible config loader, on synthetic data."""
