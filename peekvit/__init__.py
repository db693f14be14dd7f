"""Alias package so the reference's Hydra `_target_` strings (`peekvit.models.vit.VisionTransformer`, ...,
configs/model/*.yaml:1) and its import paths resolve to the MI355X-native modules unchanged."""
