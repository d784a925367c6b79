"""Checkpoint files in the reference's two formats.

The reference writes (a) ``torch.save(model.state_dict(), path)`` from its ModelCheckpoint callback
(vitm/training/callbacks.py:28-33) and (b) ``torch.save({'model_state_dict': ..., 'optimizer_state_dict': ...,
'scheduler_state_dict': ...}, path)`` at the end of training (vitm/training/train.py:245-249).  The modules of this
package have the reference's parameter names, shapes and registration order, so both load with
``strict=True`` and an optimizer state saved by the reference lines up with ``model.parameters()`` here.
Files are read with ``weights_only=True``: nothing in them is executed.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

_TRAIN_KEYS = ("model_state_dict", "optimizer_state_dict", "scheduler_state_dict")


def read_checkpoint(path: str, map_location="cpu") -> Dict[str, Any]:
    """Return ``{'model_state_dict': ..., ['optimizer_state_dict': ...], ['scheduler_state_dict': ...]}`` for
    either on-disk format."""
    obj = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: expected a dict, got {type(obj).__name__}")
    if "model_state_dict" in obj:
        return {k: obj[k] for k in _TRAIN_KEYS if k in obj}
    if not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: neither a bare state_dict nor a training checkpoint")
    return {"model_state_dict": obj}


def load_checkpoint(model: torch.nn.Module, path: str, optimizer: Optional[torch.optim.Optimizer] = None,
                    scheduler=None, strict: bool = True, map_location="cpu") -> Dict[str, Any]:
    """Load either format into ``model`` (and the optimizer / scheduler states when given and present)."""
    ck = read_checkpoint(path, map_location)
    model.load_state_dict(ck["model_state_dict"], strict=strict)
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ck:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck


def save_state_dict(model: torch.nn.Module, path: str) -> None:
    """Format (a): the bare ``state_dict`` (callbacks.py:32)."""
    torch.save(model.state_dict(), path)


def save_checkpoint(model: torch.nn.Module, path: str, optimizer: Optional[torch.optim.Optimizer] = None,
                    scheduler=None) -> None:
    """Format (b): the end-of-training dictionary (train.py:245-249)."""
    out = {"model_state_dict": model.state_dict()}
    if optimizer is not None:
        out["optimizer_state_dict"] = optimizer.state_dict()
    if scheduler is not None:
        out["scheduler_state_dict"] = scheduler.state_dict()
    torch.save(out, path)
