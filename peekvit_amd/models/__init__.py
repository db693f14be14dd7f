"""Host-side mirror of the reference's model interface for the ViT encoder hot path
(reference models/blocks.py, models/vit.py, models/rankvit.py, models/residualvit.py)."""
