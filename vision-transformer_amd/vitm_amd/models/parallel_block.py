"""ViT-22B style parallel block on MI355X.

API mirror of the reference's vitm/models/parallel_block.py -- ``LayerScale`` (:6-27),
``ParallelBlockMultiHeadAttention`` (:30-97), ``MLP`` (:100-126), ``ParallelBlock`` (:129-221) -- with the same
constructor arguments, attribute names and state_dict keys (``norm``, ``fused_proj_1``, ``attn.{q_norm,k_norm,proj}``,
``mlp.out_proj``, ``fused_proj_2``, ``layer_scale.gamma``).

The block is the same kernels as the sequential one, arranged differently:
  LayerNorm folded into two GEMMs over the rows of ``fused_proj_1`` ([q|k|v] rows, and the MLP-hidden rows with GELU
  in the epilogue) -> per-head LayerNorm of q and k -> fused attention -> ``attn.proj`` and ``mlp.out_proj`` GEMMs
  writing the two halves of one [M, 2D] buffer (the concat is free) -> ``fused_proj_2`` GEMM with bias + LayerScale
  + residual in the epilogue.
The reference's per-call ``print`` (parallel_block.py:219) is not reproduced.
"""
from __future__ import annotations

from typing import Type

import torch
import torch.nn as nn

from .. import _ops
from .models import LayerScale, _act_buffer, _as_rows, _check_gelu, _pack, _SUPPORTED_HEAD_DIMS
from .patch_embeddings import layernorm_params

__all__ = ["LayerScale", "ParallelBlockMultiHeadAttention", "MLP", "ParallelBlock"]


class ParallelBlockMultiHeadAttention(nn.Module):
    """Attention over an already projected, fused (B, N, 3*dim) QKV tensor, with LayerNorm(head_dim) on q and k."""

    def __init__(self, dim, num_heads, norm_layer: Type[nn.Module] = nn.LayerNorm):
        super().__init__()
        assert dim % num_heads == 0, "dim should be multiple of num_heads"
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.q_norm = norm_layer(self.head_dim)
        self.k_norm = norm_layer(self.head_dim)
        self.proj = nn.Linear(in_features=dim, out_features=dim)

    # ---- HIP path ------------------------------------------------------------------------------------
    def _check(self):
        if self.head_dim not in _SUPPORTED_HEAD_DIMS:
            raise NotImplementedError(f"head_dim {self.head_dim} unsupported; the fused attention kernel handles "
                                      f"head_dim in {_SUPPORTED_HEAD_DIMS}")

    def _core(self, qkv2d: torch.Tensor, o2d: torch.Tensor, B: int, N: int) -> None:
        """In place per-head LayerNorm of q and k on qkv [B*N, 3*dim], then softmax(q k^T / sqrt(hd)) v."""
        qn = layernorm_params(self.q_norm, "ParallelBlockMultiHeadAttention.q_norm")
        kn = layernorm_params(self.k_norm, "ParallelBlockMultiHeadAttention.k_norm")
        if qn is None or kn is None:
            raise NotImplementedError("q_norm / k_norm must be nn.LayerNorm")
        f = _ops.f32vec
        _ops.qk_norm(qkv2d, B * N, self.num_heads, self.head_dim, f(qn[0]), f(qn[1]), qn[2], f(kn[0]), f(kn[1]), kn[2])
        _ops.attention(qkv2d, o2d, B, N, self.num_heads, self.head_dim, self.head_dim ** -0.5)

    def forward(self, QKV: torch.Tensor) -> torch.Tensor:
        """(B, N, 3*dim) -> (B, N, dim)."""
        _ops.require_cuda(QKV, "ParallelBlockMultiHeadAttention.forward")
        self._check()
        B, N, C = QKV.shape
        assert C // 3 == self.dim, f"Inner dim of QKV {C // 3} != dim = {self.dim}"
        D = self.dim
        with torch.cuda.device(QKV.device):
            _ops.compute_dtype(QKV)
            qkv = QKV.reshape(B * N, C).contiguous()
            if qkv.data_ptr() == QKV.data_ptr():
                qkv = qkv.clone()   # q and k are normalised in place; never on the caller's tensor
            dt = qkv.dtype
            o = _act_buffer(B * N, D, QKV.device, dt)
            self._core(qkv, o, B, N)
            y = torch.empty((B * N, _ops.round_up(D, 8)), dtype=dt, device=QKV.device)
            _ops.gemm(o, _pack(self, "proj", dt, self.proj.weight, self.proj.bias), y, _ops.EPI_BIAS, B * N)
            return y[:, :D].reshape(B, N, D)


class MLP(nn.Module):
    """Activation then output projection of an already projected hidden activation (parallel_block.py:100-126)."""

    def __init__(self, mlp_hidden_features: int, out_features: int, act_layer: Type[nn.Module] = nn.GELU,
                 bias: bool = True):
        super().__init__()
        self.act_layer = act_layer()
        self.out_proj = nn.Linear(in_features=mlp_hidden_features, out_features=out_features, bias=bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _ops.require_cuda(x, "parallel_block.MLP.forward")
        _check_gelu(self.act_layer, "parallel_block.MLP")
        K, Nout = self.out_proj.in_features, self.out_proj.out_features
        with torch.cuda.device(x.device):
            xb, M = _as_rows(x, K)
            dt = xb.dtype
            h = torch.zeros_like(xb) if xb.shape[1] != K else torch.empty_like(xb)
            if K % 8 != 0:
                raise NotImplementedError("stand-alone parallel_block.MLP needs hidden features % 8 == 0")
            _ops.gelu(xb, h, M, K)
            pk = _pack(self, "out_proj", dt, self.out_proj.weight, self.out_proj.bias)
            y = torch.empty((M, pk.npad), dtype=dt, device=x.device)
            _ops.gemm(h, pk, y, _ops.EPI_BIAS, M)
            return y[:, :Nout].reshape(*x.shape[:-1], Nout)


class ParallelBlock(nn.Module):
    """``x + layer_scale(fused_proj_2(cat(attn(qkv), mlp(hidden))))`` with ``[qkv | hidden] = fused_proj_1(norm(x))``
    (parallel_block.py:193-221).  ``mlp_ratio`` is the hidden width as a fraction of ``dim`` (default 0.7)."""

    def __init__(self, dim, num_heads, mlp_ratio: float = 0.7, act_layer: Type[nn.Module] = nn.GELU,
                 mlp_bias: bool = False, vit: bool = True):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        assert dim % num_heads == 0, "dim should be multiple of num_heads"
        self.head_dim = dim // num_heads
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        out_fusion_1 = dim * 3 + self.mlp_hidden_dim
        self.norm = nn.LayerNorm(dim)
        self.fused_proj_1 = nn.Linear(in_features=dim, out_features=out_fusion_1)
        self.attn = ParallelBlockMultiHeadAttention(dim=self.dim, num_heads=num_heads)
        self.mlp = MLP(mlp_hidden_features=self.mlp_hidden_dim, out_features=dim, act_layer=act_layer, bias=mlp_bias)
        self.fused_proj_2 = nn.Linear(in_features=2 * dim, out_features=dim)
        self.layer_scale = LayerScale(dim)

    # ---- HIP path ------------------------------------------------------------------------------------
    def _run(self, x2d: torch.Tensor, B: int, N: int, ws: dict) -> None:
        """In place on the residual stream x2d ([B*N, ld], bf16 or fp32)."""
        self.attn._check()
        _check_gelu(self.mlp.act_layer, "ParallelBlock.mlp")
        D, M, dt = self.dim, B * N, x2d.dtype
        if D % 8 != 0:
            raise NotImplementedError("ParallelBlock on the HIP path needs dim % 8 == 0")
        ln = layernorm_params(self.norm, "ParallelBlock.norm")
        w1, b1 = self.fused_proj_1.weight, self.fused_proj_1.bias
        wq, wh = w1[:3 * D], w1[3 * D:]
        bq, bh = (b1[:3 * D], b1[3 * D:]) if b1 is not None else (None, None)
        if dt == torch.float32:
            src = x2d
            if ln is not None:
                _ops.layernorm(x2d, ws["xn"], M, D, _ops.f32vec(ln[0]), _ops.f32vec(ln[1]), ln[2])
                src = ws["xn"]
            _ops.gemm(src, _pack(self, "f1_qkv", dt, wq, bq), ws["qkv"], _ops.EPI_BIAS, M)
            _ops.gemm(src, _pack(self, "f1_hid", dt, wh, bh), ws["h"], _ops.EPI_BIAS_GELU, M)
        elif ln is not None:
            _ops.ln_stats(x2d, M, D, ln[2], ws["stats"])
            _ops.gemm(x2d, _pack(self, "f1_qkv", dt, wq, bq, ln=ln), ws["qkv"], _ops.EPI_LN, M, stats=ws["stats"])
            _ops.gemm(x2d, _pack(self, "f1_hid", dt, wh, bh, ln=ln), ws["h"], _ops.EPI_LN_GELU, M, stats=ws["stats"])
        else:
            _ops.gemm(x2d, _pack(self, "f1_qkv", dt, wq, bq), ws["qkv"], _ops.EPI_BIAS, M)
            _ops.gemm(x2d, _pack(self, "f1_hid", dt, wh, bh), ws["h"], _ops.EPI_BIAS_GELU, M)
        self.attn._core(ws["qkv"], ws["o"], B, N)
        cat = ws["cat"]                                   # [M, kpad(2D)]: attention half | MLP half
        _ops.gemm(ws["o"], _pack(self.attn, "proj", dt, self.attn.proj.weight, self.attn.proj.bias),
                  cat[:, :D], _ops.EPI_BIAS, M)
        _ops.gemm(ws["h"], _pack(self.mlp, "out_proj", dt, self.mlp.out_proj.weight, self.mlp.out_proj.bias),
                  cat[:, D:2 * D], _ops.EPI_BIAS, M)
        _ops.gemm(cat, _pack(self, "f2", dt, self.fused_proj_2.weight, self.fused_proj_2.bias,
                             gamma=self.layer_scale.gamma), x2d, _ops.EPI_RESIDUAL, M, resid=x2d)

    def _workspace(self, M: int, device, dtype) -> dict:
        D = self.dim
        hid_npad = _ops.round_up(self.mlp_hidden_dim, 8)
        ws = dict(qkv=torch.empty((M, 3 * D), dtype=dtype, device=device),
                  o=_act_buffer(M, D, device, dtype),
                  h=_act_buffer(M, self.mlp_hidden_dim, device, dtype, zero_pad_from=hid_npad),
                  cat=_act_buffer(M, 2 * D, device, dtype))
        if dtype == torch.float32:
            ws["xn"] = _act_buffer(M, D, device, dtype)
        else:
            ws["stats"] = torch.empty((M, 2), dtype=torch.float32, device=device)
        return ws

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, N, dim) -> (B, N, dim)."""
        _ops.require_cuda(x, "ParallelBlock.forward")
        B, N, C = x.shape
        assert C == self.dim, "inner dimension of x should be equal to the out_features of fused_linear_layer1"
        if B * N == 0:
            return x.new_empty((B, N, C), dtype=_ops.compute_dtype(x))
        with torch.cuda.device(x.device):
            xb, M = _as_rows(x, C)
            if xb.data_ptr() == x.data_ptr():
                xb = xb.clone()
            self._run(xb, B, N, self._workspace(M, x.device, xb.dtype))
            return xb[:, :C].reshape(B, N, C)
