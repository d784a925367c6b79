"""Transformer encoder modules of the ViT, computing on MI355X.

API mirror of the reference's vitm/models/models.py -- ``Attention`` (:27-95), ``MLP`` (:98-130),
``LayerScale`` (:132-139), ``Block`` (:142-211), ``VisionTransformer`` (:215-416) -- with the same
constructor arguments, attribute names and state_dict keys, so reference checkpoints load with
``load_state_dict(strict=True)``.  Every ``forward`` is a short sequence of C-ABI calls
(include/vitm_gfx950.h).  The arithmetic follows the input's dtype: bf16 tensors run the bf16 MFMA kernels
(fp32 accumulation, LayerNorm folded into the next GEMM), fp32 tensors run the fp32 kernels (fp32 MFMA,
explicit LayerNorm, erf-accurate GELU); parameters may be stored in either dtype.  The path is forward-only
(no autograd graph is recorded).

Deliberate differences from the reference's HEAD, all supersets of what can run there:
  * ``forward_heads`` iterates ``self.blocks`` (the reference calls the ModuleList, models.py:409, which raises);
  * ``embed_dim`` is forwarded to ``embed_layer`` (models.py:291-295 drops it, so only 768 works there);
  * ``norm_layer=None`` leaves ``norm_pre`` an Identity and lets the blocks use their LayerNorm default
    (the reference passes None into Block, models.py:317, and fails);
  * ``fused_attention=True`` runs the same fused kernel (the reference branch is a stub, models.py:84-85);
  * ``parallel_block=True`` builds ``ParallelBlock`` with its own signature (the reference passes Block's
    positional arguments, models.py:311-321, and fails);
  * ``qk_norm=True`` / ``qkv_bias=True`` are supported (per-head LayerNorm kernel before the attention kernel;
    the bias rides in the qkv GEMM epilogue).
"""
from __future__ import annotations

from functools import partial
from typing import Optional, Tuple, Type, Union

import torch
import torch.nn as nn

from .. import _ops
from ..utils.helper import to_2tuple
from .patch_embeddings import PatchEmbed, layernorm_params

_SUPPORTED_HEAD_DIMS = (32, 64, 96, 128)   # per-head LayerNorm (qk_norm) and the attention backward; the forward pads others
# bf16, big-tile shapes: LayerNorm row moments ride from the proj / fc2 epilogue to the qkv / fc1 epilogue
# (vitm_gemm_stats); False recomputes them with vitm_ln_stats in front of every LayerNorm-folded GEMM.
CARRY_LN_STATS = True


def _check_gelu(act: nn.Module, what: str) -> None:
    if not (isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none"):
        raise NotImplementedError(f"{what}: only nn.GELU() (exact erf form) is fused on the HIP path, got {act!r}")


def _as_rows(x: torch.Tensor, D: int) -> Tuple[torch.Tensor, int]:
    """(..., D) CUDA activation -> contiguous [M, kpad(D)] in its compute dtype (pad columns zero) and M."""
    M = x.numel() // D
    _ops.compute_dtype(x)
    xb = x.reshape(M, D).contiguous()
    kp = _ops.round_up(D, _ops.k_multiple())
    if kp != D:
        padded = torch.zeros((M, kp), dtype=x.dtype, device=x.device)
        padded[:, :D] = xb
        xb = padded
    return xb, M


def _act_buffer(M: int, width: int, device, dtype, zero_pad_from: Optional[int] = None) -> torch.Tensor:
    """[M, kpad(width)] activation buffer; zero-filled when it has pad columns a later GEMM reads."""
    kp = _ops.round_up(width, _ops.k_multiple())
    if kp != (zero_pad_from if zero_pad_from is not None else width):
        return torch.zeros((M, kp), dtype=dtype, device=device)
    return torch.empty((M, kp), dtype=dtype, device=device)


def _pack(owner, slot: str, dtype, w, b, ln=None, gamma=None) -> _ops.PackedLinear:
    """Packed form of Linear (w, b) for ``dtype`` arithmetic, with an optional folded LayerNorm (weight, bias, eps)
    in front and LayerScale vector behind; memoised on ``owner`` until a source tensor changes."""
    lw, lb = (ln[0], ln[1]) if ln is not None else (None, None)
    tag = f"{slot}{'_ln' if ln is not None else ''}{'_ls' if gamma is not None else ''}_{'f32' if dtype == torch.float32 else 'bf16'}"
    return _ops.cached_pack(owner, tag, (w, b, lw, lb, gamma),
                            lambda: _ops.pack_linear(w, b, lw, lb, out_scale=gamma, dtype=dtype))


class Attention(nn.Module):
    """Multi-head self-attention: qkv projection, softmax(q k^T / sqrt(hd)) v, output projection."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, fused_attention: bool = False,
                 qk_norm: bool = False, proj_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = self.dim // self.num_heads
        self.qkv_bias = qkv_bias
        self.fused_attention = fused_attention
        self.qk_norm = qk_norm
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=self.qkv_bias)
        self.q_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm_layer(self.head_dim) if qk_norm else nn.Identity()
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self._init_weights()

    def _init_weights(self):
        for lin in (self.qkv, self.proj):   # models.py:61-70
            nn.init.trunc_normal_(lin.weight, std=0.02)
            if lin.bias is not None:
                nn.init.zeros_(lin.bias)

    # ---- HIP path --------------------------------------------------------------------------------
    def _check(self):
        hd = self.head_dim
        if hd * self.num_heads != self.dim or hd % 8 != 0 or hd > 128:
            raise NotImplementedError(f"head_dim {self.dim}/{self.num_heads} unsupported; the fused attention kernel "
                                      f"handles head dims that are multiples of 8 up to 128")
        if not isinstance(self.q_norm, nn.Identity) and hd not in _SUPPORTED_HEAD_DIMS:
            raise NotImplementedError(f"qk_norm with head_dim {hd} unsupported; the per-head LayerNorm kernel handles "
                                      f"head_dim in {_SUPPORTED_HEAD_DIMS}")

    def _qkv_pack(self, dtype, ln=None) -> _ops.PackedLinear:
        return _pack(self, "qkv", dtype, self.qkv.weight, self.qkv.bias, ln=ln)

    def _proj_pack(self, dtype, gamma=None) -> _ops.PackedLinear:
        return _pack(self, "proj", dtype, self.proj.weight, self.proj.bias, gamma=gamma)

    def _core(self, qkv2d: torch.Tensor, o2d: torch.Tensor, B: int, N: int) -> None:
        """qkv [B*N, 3C] -> attention output [B*N, C]; with qk_norm the per-head LayerNorm of q and k
        (models.py:82) runs in place on qkv first."""
        qn = layernorm_params(self.q_norm, "Attention.q_norm")
        kn = layernorm_params(self.k_norm, "Attention.k_norm")
        if (qn is None) != (kn is None):
            raise NotImplementedError("q_norm and k_norm must both be LayerNorm or both Identity")
        if qn is not None:
            f = _ops.f32vec
            _ops.qk_norm(qkv2d, B * N, self.num_heads, self.head_dim, f(qn[0]), f(qn[1]), qn[2],
                         f(kn[0]), f(kn[1]), kn[2])
        _ops.attention(qkv2d, o2d, B, N, self.num_heads, self.head_dim, self.scale)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, N, C) -> (B, N, C)."""
        _ops.require_cuda(x, "Attention.forward")
        self._check()
        B, N, C = x.shape
        if B * N == 0:                            # an empty batch (or no tokens) is an empty result, as in the reference
            return x.new_empty((B, N, C), dtype=_ops.compute_dtype(x))
        with torch.cuda.device(x.device):
            xb, M = _as_rows(x, C)
            dt = xb.dtype
            qkv = torch.empty((M, 3 * C), dtype=dt, device=x.device)
            _ops.gemm(xb, self._qkv_pack(dt), qkv, _ops.EPI_BIAS, M)
            o = _act_buffer(M, C, x.device, dt)
            self._core(qkv, o, B, N)
            y = torch.empty((M, _ops.round_up(C, 8)), dtype=dt, device=x.device)
            _ops.gemm(o, self._proj_pack(dt), y, _ops.EPI_BIAS, M)
            return y[:, :C].reshape(B, N, C)


class MLP(nn.Module):
    """Two-layer feed-forward: linear -> GELU -> linear (models.py:98-130).

    As in the reference, ``out_features`` is ignored (the output width is ``in_features``), ``bias`` may be a
    bool or a pair, and ``norm_layer`` / ``vit`` are accepted and unused.  ``use_conv=True`` builds the 1x1
    Conv2d layers (models.py:119) and takes (B, C, H, W) input: a 1x1 convolution is the same GEMM on
    ``weight.view(out, in)`` over the pixels as rows.  The exact GELU is fused into the first GEMM's epilogue; ReLU,
    tanh-GELU, SiLU, Sigmoid and Tanh run as one elementwise pass between the two GEMMs (``vitm_activation``).
    """

    def __init__(self, in_features, hidden_features: int = None, out_features: int = None, use_conv: bool = False,
                 bias: bool = False, act_layer: Type[nn.Module] = nn.GELU, norm_layer=None, vit: bool = False):
        super().__init__()
        self.vit = vit
        self.in_features = in_features
        self.hidden_features = hidden_features if hidden_features is not None else self.in_features * 4
        self.out_features = self.in_features or out_features
        self.bias = to_2tuple(bias)
        self.use_conv = use_conv
        self.linear_layer = partial(nn.Conv2d, kernel_size=1) if use_conv else nn.Linear
        self.linear_layer1 = self.linear_layer(self.in_features, self.hidden_features, bias=self.bias[0])
        self.ac1 = act_layer()
        self.linear_layer2 = self.linear_layer(self.hidden_features, self.out_features, bias=self.bias[1])

    # ---- HIP path --------------------------------------------------------------------------------
    def _fc1_pack(self, dtype, ln=None) -> _ops.PackedLinear:
        return _pack(self, "fc1", dtype, self.linear_layer1.weight, self.linear_layer1.bias, ln=ln)

    def _fc2_pack(self, dtype, gamma=None) -> _ops.PackedLinear:
        return _pack(self, "fc2", dtype, self.linear_layer2.weight, self.linear_layer2.bias, gamma=gamma)

    def _act_kind(self) -> int:
        kind = _ops.activation_kind(self.ac1)
        if kind is None:
            raise NotImplementedError(f"MLP: no kernel for activation {self.ac1!r} (nn.GELU exact or tanh, nn.ReLU, "
                                      f"nn.SiLU, nn.Sigmoid, nn.Tanh)")
        return kind

    def _check(self):
        self._act_kind()

    def _fc1(self, xb: torch.Tensor, pk, h: torch.Tensor, M: int, ln_epilogue: bool = False, **stats) -> None:
        """First GEMM + activation into h: the exact GELU in the GEMM's epilogue, any other activation as a pass over h."""
        kind = self._act_kind()
        fused = kind == _ops._lib.ACT_GELU_ERF
        if ln_epilogue:
            epi = _ops.EPI_LN_GELU if fused else _ops.EPI_LN
        else:
            epi = _ops.EPI_BIAS_GELU if fused else _ops.EPI_BIAS
        _ops.gemm(xb, pk, h, epi, M, **stats)
        if not fused:
            _ops.activation(h, h, M, pk.npad, kind)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(..., in_features) -> (..., in_features); with ``use_conv=True``: (B, C, H, W) -> (B, C, H, W)."""
        _ops.require_cuda(x, "MLP.forward")
        self._check()
        C = self.in_features
        if self.use_conv:
            if x.dim() != 4 or x.shape[1] != C:
                raise RuntimeError(f"MLP(use_conv=True) takes (B, {C}, H, W) input, got {tuple(x.shape)}")
            rows = x.permute(0, 2, 3, 1)          # pixels as rows (a layout copy, no arithmetic)
            lead = rows.shape[:-1]
        else:
            rows, lead = x, x.shape[:-1]
        if x.numel() == 0:
            return x.new_empty(x.shape, dtype=_ops.compute_dtype(x))
        with torch.cuda.device(x.device):
            xb, M = _as_rows(rows, C)
            dt = xb.dtype
            p1, p2 = self._fc1_pack(dt), self._fc2_pack(dt)
            h = _act_buffer(M, self.hidden_features, x.device, dt, zero_pad_from=p1.npad)
            self._fc1(xb, p1, h, M)
            y = torch.empty((M, p2.npad), dtype=dt, device=x.device)
            _ops.gemm(h, p2, y, _ops.EPI_BIAS, M)
            y = y[:, :self.out_features].reshape(*lead, self.out_features)
            return y.permute(0, 3, 1, 2).contiguous() if self.use_conv else y


Mlp = MLP  # timm spelling


class LayerScale(nn.Module):
    """Per-channel learnable scale ``x * gamma`` (models.py:132-139)."""

    def __init__(self, dim: int, init_values: float = 1e-5, inplace=False):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim) * init_values)
        self.inplace = inplace

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _ops.require_cuda(x, "LayerScale.forward")
        D = self.gamma.numel()
        if D % 8 != 0:
            raise NotImplementedError("LayerScale on the HIP path needs dim % 8 == 0")
        if x.numel() == 0:
            return x if self.inplace else x.new_empty(x.shape, dtype=_ops.compute_dtype(x))
        with torch.cuda.device(x.device):
            M = x.numel() // D
            g = _ops.f32vec(self.gamma)
            _ops.compute_dtype(x)
            if self.inplace and x.is_contiguous():
                x2 = x.view(M, D)
                _ops.scale_channels(x2, x2, M, D, g)
                return x
            xb = x.reshape(M, D).contiguous()
            out = torch.empty_like(xb)
            _ops.scale_channels(xb, out, M, D, g)
            y = out.reshape(x.shape)
            if self.inplace:
                x.copy_(y)
                return x
            return y


class Block(nn.Module):
    """Pre-norm transformer block: ``x += ls1(attn(norm1(x)))``; ``x += ls2(mlp(norm2(x)))`` (models.py:208-211).

    Attention is built with the reference's defaults (no qkv bias, proj bias, no qk-norm, models.py:190-194) and
    the MLP with biases on both layers and ``hidden = int(dim * mlp_ratio)`` (models.py:198-205).
    On the bf16 path at large batch a block is five launches: LN-folded qkv GEMM, fused attention, proj GEMM with
    bias + LayerScale + residual, LN-folded fc1 GEMM with GELU, fc2 GEMM with bias + LayerScale + residual; the proj
    and fc2 epilogues also leave the row moments that the next LayerNorm-folded GEMM turns into mean and rstd.  At
    small batch (128x128-tile GEMMs) a LayerNorm-statistics launch precedes each folded GEMM: seven launches.  The
    fp32 path has those seven with a full LayerNorm in place of the statistics pass (the fp32 GEMM does not fold it).
    """

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float, norm_layer: Type[nn.Module] = nn.LayerNorm,
                 act_layer: Type[nn.Module] = nn.GELU, mlp_layer=MLP, vit: bool = False):
        super().__init__()
        self.dim = dim
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim=dim, num_heads=num_heads, norm_layer=norm_layer)
        self.ls1 = LayerScale(dim=dim)
        self.norm2 = norm_layer(dim)
        self.mlp = MLP(in_features=dim, hidden_features=int(dim * mlp_ratio), bias=True, norm_layer=norm_layer,
                       act_layer=act_layer, vit=vit)
        self.ls2 = LayerScale(dim=dim)

    # ---- HIP path --------------------------------------------------------------------------------
    def _run(self, x2d: torch.Tensor, B: int, N: int, ws: dict) -> None:
        """In place on the residual stream x2d ([B*N, ld], bf16 or fp32); ``ws`` holds the reusable work buffers."""
        attn, mlp = self.attn, self.mlp
        attn._check()
        mlp._check()
        D, M, dt = self.dim, B * N, x2d.dtype
        ln1 = layernorm_params(self.norm1, "Block.norm1")
        ln2 = layernorm_params(self.norm2, "Block.norm2")
        f32 = dt == torch.float32

        def normed(ln):   # fp32 path: explicit LayerNorm into the work buffer
            _ops.layernorm(x2d, ws["xn"], M, D, _ops.f32vec(ln[0]), _ops.f32vec(ln[1]), ln[2])
            return ws["xn"]

        # bf16 at large batch: the proj / fc2 epilogues also leave the row moments of the stream they write, so the
        # LayerNorm in front of the next GEMM needs no pass of its own over x (ws["carried"]; ws["carried_valid"]
        # says whether it describes the current x2d -- the first block of a forward starts from vitm_ln_stats)
        proj_pk = attn._proj_pack(dt, self.ls1.gamma)
        fc2_pk = mlp._fc2_pack(dt, self.ls2.gamma)
        carry = False
        if CARRY_LN_STATS and not f32 and ln1 is not None and ln2 is not None and D <= 256 * _ops._lib.STATS_PARTS:
            memo = self.__dict__.setdefault("_vitm_carry", {})   # the route depends on the shapes only
            carry = memo.get(M)
            if carry is None:
                carry = memo[M] = all(_ops.uses_big_tile(M, pk) for pk in
                                      (proj_pk, fc2_pk, attn._qkv_pack(dt, ln1), mlp._fc1_pack(dt, ln2)))
        if carry and "carried" not in ws:
            ws["carried"] = _ops.new_carried_stats(M, x2d.device)
            ws["carried_valid"] = False
        # attention branch
        if ln1 is None:
            _ops.gemm(x2d, attn._qkv_pack(dt), ws["qkv"], _ops.EPI_BIAS, M)
        elif f32:
            _ops.gemm(normed(ln1), attn._qkv_pack(dt), ws["qkv"], _ops.EPI_BIAS, M)
        elif carry and ws["carried_valid"]:
            _ops.gemm(x2d, attn._qkv_pack(dt, ln1), ws["qkv"], _ops.EPI_LN, M, stats=ws["carried"], stats_partial=True,
                      ln_eps=ln1[2])
        else:
            if not ws.pop("stats_valid", False):   # True: the producer of x2d (embed_finalize) already left them
                _ops.ln_stats(x2d, M, D, ln1[2], ws["stats"])
            _ops.gemm(x2d, attn._qkv_pack(dt, ln1), ws["qkv"], _ops.EPI_LN, M, stats=ws["stats"])
        attn._core(ws["qkv"], ws["o"], B, N)
        _ops.gemm(ws["o"], proj_pk, x2d, _ops.EPI_RESIDUAL, M, resid=x2d, stats_out=ws["carried"] if carry else None)
        # MLP branch
        if ln2 is None:
            mlp._fc1(x2d, mlp._fc1_pack(dt), ws["h"], M)
        elif f32:
            mlp._fc1(normed(ln2), mlp._fc1_pack(dt), ws["h"], M)
        elif carry:
            mlp._fc1(x2d, mlp._fc1_pack(dt, ln2), ws["h"], M, ln_epilogue=True, stats=ws["carried"], stats_partial=True,
                     ln_eps=ln2[2])
        else:
            _ops.ln_stats(x2d, M, D, ln2[2], ws["stats"])
            mlp._fc1(x2d, mlp._fc1_pack(dt, ln2), ws["h"], M, ln_epilogue=True, stats=ws["stats"])
        _ops.gemm(ws["h"], fc2_pk, x2d, _ops.EPI_RESIDUAL, M, resid=x2d, stats_out=ws["carried"] if carry else None)
        ws["carried_valid"] = carry

    def _workspace(self, M: int, device, dtype) -> dict:
        D = self.dim
        hid_npad = _ops.round_up(self.mlp.hidden_features, 8)
        ws = dict(qkv=torch.empty((M, 3 * D), dtype=dtype, device=device),
                  o=_act_buffer(M, D, device, dtype),
                  h=_act_buffer(M, self.mlp.hidden_features, device, dtype, zero_pad_from=hid_npad))
        if dtype == torch.float32:
            ws["xn"] = _act_buffer(M, D, device, dtype)
        else:
            ws["stats"] = torch.empty((M, 2), dtype=torch.float32, device=device)
        return ws

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, N, D) -> (B, N, D)."""
        _ops.require_cuda(x, "Block.forward")
        B, N, D = x.shape
        if D != self.dim:
            raise RuntimeError(f"Block(dim={self.dim}) got an input with last dim {D}")
        if B * N == 0:
            return x.new_empty((B, N, D), dtype=_ops.compute_dtype(x))
        with torch.cuda.device(x.device):
            xb, M = _as_rows(x, D)
            if xb.data_ptr() == x.data_ptr():
                xb = xb.clone()  # the block updates its residual stream in place; never the caller's tensor
            self._run(xb, B, N, self._workspace(M, x.device, xb.dtype))
            return xb[:, :D].reshape(B, N, D)


class VisionTransformer(nn.Module):
    """timm-style ViT encoder: patch embedding, class token, ``norm_pre`` (applied BEFORE the positional
    embedding is added, models.py:359-360), ``depth`` blocks, ``norm_final``; returns all tokens (B, N, D).

    Constructor arguments are those of models.py:261-281 (``parallel_block=True`` selects the ViT-22B block of
    parallel_block.py).  The DTensor helpers ``parallelize_blocks`` / ``spawn_parallel`` are outside this forward path.
    """

    def __init__(self, image_size: Union[int, Tuple[int, int]], patch_size: int, num_classes: int, num_heads: int,
                 mlp_ratio: float, depth: int = 12, embed_dim: int = 768, block_fn=Block,
                 norm_layer: Optional[nn.Module] = None, mlp_layer=MLP, pos_embed: str = "learn",
                 embed_norm_layer: Optional[nn.Module] = None, embed_layer: Optional[nn.Module] = PatchEmbed,
                 class_token: bool = True, final_norm_layer: Optional[nn.Module] = None,
                 act_layer: Optional[nn.Module] = nn.GELU, parallel_block: bool = False,
                 shard_along_dims: tuple = None):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_classes = num_classes
        self.image_size = image_size
        self.patch_size = patch_size
        embed_args = {}
        if embed_norm_layer is not None:
            embed_args["norm_layer"] = embed_norm_layer
        try:
            self.patch_embed = embed_layer(self.image_size, self.patch_size, embed_dim=embed_dim, **embed_args)
        except TypeError:   # an embed_layer that does not take embed_dim: the reference's call form
            self.patch_embed = embed_layer(self.image_size, self.patch_size, **embed_args)
        embed_len = self.patch_embed.num_patches

        # as in the reference (models.py:299) the token exists for any non-None ``class_token`` value
        self.class_token = None
        if class_token is not None:
            self.class_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            nn.init.trunc_normal_(self.class_token, std=0.02)
        pos_embed_len = embed_len + 1 if self.class_token is not None else embed_len
        self.pos_embed = None
        if pos_embed == "learn":
            self.pos_embed = nn.Parameter(torch.zeros(1, pos_embed_len, embed_dim))
            nn.init.trunc_normal_(self.pos_embed, std=0.02)

        self.norm_pre = norm_layer(embed_dim) if norm_layer else nn.Identity()
        block_norm = norm_layer if norm_layer else nn.LayerNorm
        if parallel_block:
            # the reference passes Block's positional arguments to ParallelBlock (models.py:311-321), which puts the
            # norm layer in the activation slot and fails; this is the call its signature (parallel_block.py:172-180)
            # asks for
            from .parallel_block import ParallelBlock
            self.blocks = nn.ModuleList([
                ParallelBlock(embed_dim, num_heads, mlp_ratio, act_layer=act_layer, vit=True) for _ in range(depth)])
        else:
            self.blocks = nn.ModuleList([
                block_fn(embed_dim, num_heads, mlp_ratio, block_norm, act_layer, vit=True) for _ in range(depth)])
        self.norm_final = final_norm_layer(embed_dim) if final_norm_layer else nn.Identity()
        self.feature_info = [dict(module=f"block_{i}", num_chs=embed_dim) for i in range(depth)]
        self.parallel_block = parallel_block
        self.shard_along_dims = shard_along_dims
        self._init_weights()

    def _init_weights(self):
        """Same initialisation law as models.py:338-350."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def parallelize_blocks(self, device_id, world_size):
        raise NotImplementedError("DTensor TP/FSDP of ParallelBlock (models.py:363-397) is outside this forward path; "
                                  "inference is batch-sharded across GPUs with one replica per GPU")

    def spawn_parallel(self, device_id: int, world_size: int, path_to_config: str = "models/sv22b.yaml"):
        pass  # a no-op in the reference as well (models.py:399-406)

    # ---- HIP path --------------------------------------------------------------------------------
    def _features2d(self, x: torch.Tensor, stats: Optional[torch.Tensor] = None, stats_eps: float = 0.0):
        """forward_features on the residual-stream buffer: returns (x2d [B*N, ld] in x's dtype, B, N, D).  ``stats``
        ([B*N, 2] fp32) also receives the LayerNorm statistics of the finished rows (eps ``stats_eps``)."""
        pe = self.patch_embed
        native = isinstance(pe, PatchEmbed)
        if native and not pe.flatten:
            raise NotImplementedError("VisionTransformer needs (B, N, D) tokens: PatchEmbed(flatten=True)")
        if self.pos_embed is None:
            raise TypeError("pos_embed is None: the reference adds it unconditionally (models.py:360)")
        B = x.shape[0]
        D = self.embed_dim
        has_cls = self.class_token is not None
        tokens = None
        if native:
            Np = pe.num_patches
        else:
            # an embed_layer of the caller's own (the reference instantiates and calls whatever class it is given,
            # models.py:275,352): its forward produces the tokens, everything behind it runs here
            _ops.require_cuda(x, "VisionTransformer.forward")
            tokens = pe(x)
            if tokens.dim() != 3 or tokens.shape[0] != B or tokens.shape[2] != D:
                raise RuntimeError(f"embed_layer returned {tuple(tokens.shape)}, expected (B, N, {D}) tokens")
            Np = tokens.shape[1]
        N = Np + (1 if has_cls else 0)
        if self.pos_embed.shape[-2] != N:
            raise RuntimeError(f"pos_embed holds {self.pos_embed.shape[-2]} positions, the embedding produced {N} tokens")
        ld = _ops.round_up(D, _ops.k_multiple())
        alloc = torch.zeros if ld != D else torch.empty
        x2d = alloc((B * N, ld), dtype=_ops.compute_dtype(x), device=x.device)
        if native:
            pe._tokens_into(x, x2d, cls_slot=has_cls)
            en = layernorm_params(pe.norm_layer, "PatchEmbed.norm_layer")
        else:
            x2d.view(B, N, ld)[:, (1 if has_cls else 0):, :D].copy_(tokens)
            en = None                     # whatever norm the foreign layer has is part of its forward
        npre = layernorm_params(self.norm_pre, "VisionTransformer.norm_pre")
        f = _ops.f32vec
        _ops.embed_finalize(
            x2d, B, N, D, f(self.class_token).reshape(-1) if has_cls else None, f(self.pos_embed).reshape(N, D),
            f(en[0]) if en else None, f(en[1]) if en else None, en[2] if en else 0.0,
            f(npre[0]) if npre else None, f(npre[1]) if npre else None, npre[2] if npre else 0.0,
            stats=stats, stats_eps=stats_eps)
        return x2d, B, N, D

    def _run_blocks(self, x: torch.Tensor):
        """Everything up to (not including) ``norm_final``; returns (x2d, B, N, D)."""
        ws = None
        first = self.blocks[0] if len(self.blocks) else None
        if type(first) is Block and _ops.compute_dtype(x) == torch.bfloat16 and isinstance(self.patch_embed, PatchEmbed):
            # bf16: the kernel that finishes the embedding also leaves the row statistics norm1 of block 0 needs
            ln1 = layernorm_params(first.norm1, "Block.norm1")
            if ln1 is not None:
                Ntok = self.patch_embed.num_patches + (1 if self.class_token is not None else 0)
                ws = first._workspace(x.shape[0] * Ntok, x.device, torch.bfloat16)
                x2d, B, N, D = self._features2d(x, stats=ws["stats"], stats_eps=ln1[2])
                ws["stats_valid"] = True
        if ws is None:
            x2d, B, N, D = self._features2d(x)
        for blk in self.blocks:
            if not hasattr(blk, "_run"):
                raise NotImplementedError(f"block_fn {type(blk).__name__} has no HIP forward")
            if ws is None:
                ws = blk._workspace(B * N, x.device, x2d.dtype)
            blk._run(x2d, B, N, ws)
        return x2d, B, N, D

    def _final_norm(self, x2d: torch.Tensor, B: int, N: int, D: int, like: torch.Tensor) -> torch.Tensor:
        ln = layernorm_params(self.norm_final, "VisionTransformer.norm_final")
        if ln is None:
            return x2d[:, :D].reshape(B, N, D)
        out = torch.empty((B * N, D), dtype=x2d.dtype, device=x2d.device)
        _ops.layernorm(x2d, out, B * N, D, _ops.f32vec(ln[0]), _ops.f32vec(ln[1]), ln[2])
        return out.reshape(B, N, D)

    def _empty_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """The (0, N, D) result of an empty image batch (after the embedding layer's own shape checks)."""
        pe = self.patch_embed
        Np = pe(x).shape[1] if not isinstance(pe, PatchEmbed) else (pe._check_image(x), pe.num_patches)[1]
        N = Np + (1 if self.class_token is not None else 0)
        return x.new_empty((0, N, self.embed_dim), dtype=_ops.compute_dtype(x))

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        """Images (B, C, H, W) -> tokens (B, N, D) entering the first block (models.py:352-361)."""
        _ops.require_cuda(x, "VisionTransformer.forward_features")
        if x.shape[0] == 0:
            return self._empty_tokens(x)
        with torch.cuda.device(x.device):
            x2d, B, N, D = self._features2d(x)
            return x2d[:, :D].reshape(B, N, D)

    def forward_heads(self, x: torch.Tensor) -> torch.Tensor:
        """Tokens (B, N, D) -> blocks in order -> ``norm_final`` (models.py:408-411)."""
        _ops.require_cuda(x, "VisionTransformer.forward_heads")
        B, N, D = x.shape
        if B * N == 0:
            return x.new_empty((B, N, D), dtype=_ops.compute_dtype(x))
        with torch.cuda.device(x.device):
            xb, M = _as_rows(x, D)
            if xb.data_ptr() == x.data_ptr():
                xb = xb.clone()
            ws = None
            for blk in self.blocks:
                if not hasattr(blk, "_run"):
                    raise NotImplementedError(f"block_fn {type(blk).__name__} has no HIP forward")
                if ws is None:
                    ws = blk._workspace(M, x.device, xb.dtype)
                blk._run(xb, B, N, ws)
            return self._final_norm(xb, B, N, D, x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Images (B, C, H, W) -> final-normed token embeddings (B, N, D) (models.py:413-416)."""
        _ops.require_cuda(x, "VisionTransformer.forward")
        if x.shape[0] == 0:
            return self._empty_tokens(x)
        with torch.cuda.device(x.device):
            x2d, B, N, D = self._run_blocks(x)
            return self._final_norm(x2d, B, N, D, x)
