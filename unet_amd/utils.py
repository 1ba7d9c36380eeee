"""Small host helpers mirrored from the reference (solpex/utils.py:11-34)."""
from __future__ import annotations

import torch


def pick_device():
    """Reference: solpex/utils.py:18-23.  On PyTorch-ROCm `torch.cuda.is_available()` is true on an MI355X."""
    if torch.cuda.is_available():
        return "cuda"
    if getattr(torch.backends, "mps", None) is not None and torch.backends.mps.is_available():
        return "mps"
    return "cpu"


def sample_from_loader(loader, k=0):
    """k-th item of the first batch of a loader as a batch of one (reference: solpex/utils.py:26-34)."""
    batch = next(iter(loader))
    return {key: (v[k:k + 1] if torch.is_tensor(v) else v) for key, v in batch.items()}
