"""Image -> patch tokens on MI355X.

API mirror of the reference's ``PatchEmbed`` (vitm/models/patch_embeddings.py:14-121): same constructor,
attributes and state_dict keys (``projection.weight`` of shape (D, C, ph, pw), optional ``projection.bias``,
optional ``norm_layer.{weight,bias}``).  The strided convolution runs as an im2col-fused MFMA GEMM
(``vitm_patch_embed``); the reference's per-call ``print`` (patch_embeddings.py:112) is not reproduced.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import _ops
from ..utils.helper import Format, nchw_to, to_2tuple


def layernorm_params(mod: nn.Module, what: str):
    """(weight, bias, eps) of an affine nn.LayerNorm, or None for nn.Identity; anything else is refused."""
    if isinstance(mod, nn.Identity):
        return None
    if isinstance(mod, nn.LayerNorm) and len(mod.normalized_shape) == 1:
        if mod.elementwise_affine and mod.bias is not None:
            return mod.weight, mod.bias, mod.eps
        # no affine (or no bias): unit weight / zero bias stand in, kept on the module.  They live where the weight lives
        # or, for a module without parameters, on the current device (the callers run under torch.cuda.device(x.device)),
        # and are rebuilt when that changes (model.to(another GPU))
        dev = mod.weight.device if mod.weight is not None else torch.device("cuda", torch.cuda.current_device())
        unit = getattr(mod, "_vitm_unit", None)
        if unit is None or unit[0].device != dev:
            unit = (torch.ones(mod.normalized_shape[0], device=dev), torch.zeros(mod.normalized_shape[0], device=dev))
            object.__setattr__(mod, "_vitm_unit", unit)
        return (mod.weight if mod.weight is not None else unit[0]), unit[1], mod.eps
    raise NotImplementedError(f"{what}: only nn.LayerNorm (affine, last dim) or nn.Identity run on the HIP path, "
                              f"got {type(mod).__name__}")


class PatchEmbed(nn.Module):
    """Split an image into non-overlapping patches and project each to ``embed_dim``.

    Args mirror patch_embeddings.py:44-54: ``image_size`` (int or (H, W)), ``patch_size``, ``in_channels``,
    ``embed_dim``, ``norm_layer`` (callable building a norm over embed_dim, or None), ``flatten``
    ((B, N, D) output when True), ``bias`` (projection bias, default off), ``output_fmt`` (layout used
    when ``flatten=False``).
    """

    def __init__(self, image_size: Union[int, Tuple[int, int]] = 224, patch_size: int = 16, in_channels: int = 3,
                 embed_dim: int = 768, norm_layer: Optional[Callable] = None, flatten: bool = True,
                 bias: bool = False, output_fmt: Optional[Format] = None):
        super().__init__()
        self.output_fmt = output_fmt
        self.patch_size = to_2tuple(patch_size)
        self.image_size, self.grid_size, self.num_patches = self._init_img_size(image_size)
        self.flatten = flatten
        self.projection = nn.Conv2d(in_channels, embed_dim, kernel_size=self.patch_size, stride=self.patch_size,
                                    bias=bias)
        nn.init.kaiming_normal_(self.projection.weight, mode="fan_in", nonlinearity="relu")
        self.norm_layer = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def _init_img_size(self, image_size):
        """Resolved (H, W), patches per side [gh, gw] and their product (patch_embeddings.py:71-87)."""
        image_size = to_2tuple(image_size)
        grid_size = [s // p for s, p in zip(image_size, self.patch_size)]
        return image_size, grid_size, grid_size[0] * grid_size[1]

    # ---- HIP path --------------------------------------------------------------------------------
    def _packed_projection(self, dtype) -> _ops.PackedLinear:
        w, b = self.projection.weight, self.projection.bias
        slot = "projection_f32" if dtype == torch.float32 else "projection_bf16"
        return _ops.cached_pack(self, slot, (w, b), lambda: _ops.pack_linear(w, b, dtype=dtype))

    def _check_image(self, x: torch.Tensor) -> None:
        B, C, H, W = x.shape
        assert self.image_size[0] == H, "Input tensor's height does not match the model"
        assert self.image_size[1] == W, "Input tensor's width does not match the model"
        if C != self.projection.in_channels:
            raise RuntimeError(f"expected {self.projection.in_channels} input channels, got {C}")

    def _tokens_into(self, x: torch.Tensor, tokens2d: torch.Tensor, cls_slot: bool) -> None:
        """Write the projected patch tokens of image batch ``x`` into ``tokens2d`` (rows (b, n), in x's dtype)."""
        self._check_image(x)
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"unsupported image dtype {x.dtype} (bf16 or fp32)")
        _ops.patch_embed(x.contiguous(), self._packed_projection(tokens2d.dtype), tokens2d, self.patch_size,
                         self.projection.out_channels, cls_slot)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, H, W) -> (B, N, D) when ``flatten`` else (B, D, gh, gw) / ``output_fmt`` layout."""
        _ops.require_cuda(x, "PatchEmbed.forward")
        B, C, H, W = x.shape
        assert self.image_size[0] == H, "Input tensor's height does not match the model"
        assert self.image_size[1] == W, "Input tensor's width does not match the model"
        D, Np = self.projection.out_channels, self.num_patches
        if B == 0:                                # an empty batch is an empty result, as in the reference
            self._check_image(x)
            if self.flatten:
                return x.new_empty((0, Np, D), dtype=_ops.compute_dtype(x))
            y = x.new_empty((0, D, self.grid_size[0], self.grid_size[1]), dtype=_ops.compute_dtype(x))
            return y if self.output_fmt == "NCHW" else nchw_to(y, self.output_fmt).contiguous()
        with torch.cuda.device(x.device):
            ld = _ops.round_up(D, 8)
            tok = torch.empty((B * Np, ld), dtype=_ops.compute_dtype(x), device=x.device)
            self._tokens_into(x, tok, cls_slot=False)
            if self.flatten:
                ln = layernorm_params(self.norm_layer, "PatchEmbed.norm_layer")
                if ln is not None:
                    out = torch.empty_like(tok)
                    _ops.layernorm(tok, out, B * Np, D, _ops.f32vec(ln[0]), _ops.f32vec(ln[1]), ln[2])
                    tok = out
                return tok[:, :D].reshape(B, Np, D)
            # flatten=False: the tokens viewed back as NCHW (a layout view, no arithmetic)
            y = tok[:, :D].reshape(B, self.grid_size[0], self.grid_size[1], D).permute(0, 3, 1, 2)
            if self.output_fmt != "NCHW":
                y = nchw_to(y, self.output_fmt)
            ln = layernorm_params(self.norm_layer, "PatchEmbed.norm_layer")
            if ln is not None:
                if y.shape[-1] != D:
                    raise RuntimeError(f"norm_layer over {D} features cannot act on a last dim of {y.shape[-1]}")
                y2 = y.reshape(-1, D).contiguous()
                out = torch.empty_like(y2)
                _ops.layernorm(y2, out, y2.shape[0], D, _ops.f32vec(ln[0]), _ops.f32vec(ln[1]), ln[2])
                y = out.reshape(y.shape)
            return y.contiguous()
