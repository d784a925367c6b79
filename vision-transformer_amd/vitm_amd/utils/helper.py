"""Layout helpers used by PatchEmbed.

Behavioural mirror of the three helpers the reference's hot path imports from
vitm/utils/helper.py (``to_2tuple`` :14-21, ``Format`` :25-29, ``nchw_to`` :31-38); the rest of that
file (profiling/config plumbing) is outside the forward path and is not provided.
"""
from __future__ import annotations

from enum import Enum
from typing import Any, Tuple

import torch


def to_2tuple(value: Any) -> Tuple[Any, ...]:
    """``7 -> (7, 7)``; any non-string iterable is returned as a tuple of its items."""
    if isinstance(value, str):
        return (value, value)
    try:
        return tuple(value)
    except TypeError:
        return (value, value)


class Format(str, Enum):
    """Tensor layouts PatchEmbed can emit when ``flatten=False``."""
    NCHW = "NCHW"
    NHWC = "NHWC"
    NCL = "NCL"
    NLC = "NLC"


def nchw_to(x: torch.Tensor, fmt: Format) -> torch.Tensor:
    """View an NCHW tensor in another layout (no copy); NCHW and unknown names pass through."""
    name = fmt.value if isinstance(fmt, Format) else fmt
    if name == "NHWC":
        return x.permute(0, 2, 3, 1)
    if name == "NLC":
        return x.flatten(2).transpose(1, 2)
    if name == "NCL":
        return x.flatten(2)
    return x
