"""The training step of ``VITC`` on MI355X: fp32 (the reference's precision) or bf16 mixed precision.

Counterpart of the loop body of the reference's vitm/training/train.py:215-224::

    optimizer.zero_grad(); outputs = vitc(inputs); loss = criterion(outputs, labels); loss.backward()
    torch.nn.utils.clip_grad_norm_(vitc.parameters(), max_norm=1.0); optimizer.step(); scheduler.step()

with every tensor operation in hand-written HIP kernels: the forward keeps the activations the backward needs, the
backward is written out by hand (GEMMs of the forward kernel family on transposed operands, fp32 attention backward,
LayerNorm / GELU backward, deterministic column-sum reductions), the loss, the gradient clip and the AdamW update are
kernels too.  The reference trains in fp32 (no autocast), and fp32 images take that path.  bf16 images (or
``compute_dtype=torch.bfloat16``) take the mixed-precision path: parameters, their gradients, the optimizer state,
LayerNorm / softmax statistics, every reduction and the loss stay fp32; activations, their gradients and the GEMM
operands are bf16 on the bf16 matrix cores with fp32 accumulation (weight gradients are stored in fp32 straight from
the accumulators).  PyTorch supplies memory, the autograd hook-up (one ``autograd.Function`` for the whole model, one
for the loss) and, for data parallelism, the RCCL all-reduce of the flat gradient buffer.

Supported: ``VITC(VisionTransformer)`` built from this package's sequential ``Block`` (the configuration the
reference's training script builds, train.py:167-177; LayerNorm or Identity norms, ``qk_norm`` / ``qkv_bias`` attention)
or from ``ParallelBlock`` (``VisionTransformer(parallel_block=True)``), with a learnable position embedding and GELU.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _ops
from . import _train_ops as T
from .models.models import Block, VisionTransformer, _check_gelu
from .models.parallel_block import ParallelBlock
from .models.patch_embeddings import PatchEmbed, layernorm_params

F32 = torch.float32


# ----------------------------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------------------------
def _ld(D: int) -> int:
    return _ops.round_up(D, _ops.k_multiple())


def _buf(M: int, width: int, device, dt: torch.dtype = F32) -> torch.Tensor:
    """[M, roundup(width, 64)] in the compute dtype; zero-filled when it has pad columns (they feed GEMMs as K padding)."""
    w = _ld(width)
    return torch.zeros((M, w), dtype=dt, device=device) if w != width else torch.empty((M, w), dtype=dt, device=device)


def _pk(w: torch.Tensor, b: Optional[torch.Tensor], dt: torch.dtype = F32):
    """The fp32 master weight packed in the compute dtype for this step (weights change every step: no caching)."""
    return _ops.pack_linear(w, b, dtype=dt)


def _ln(mod: nn.Module, what: str):
    p = layernorm_params(mod, what)
    if p is None:
        return None
    return _ops.f32vec(p[0]), _ops.f32vec(p[1]), p[2], p[0], p[1]   # (w32, b32, eps, weight Parameter, bias Parameter)


def supported(model) -> Optional[str]:
    """None when the training path can run this model, else the reason."""
    vt = getattr(model, "vit_model", None)
    if not isinstance(vt, VisionTransformer):
        return "VITC.vit_model must be this package's VisionTransformer"
    if not isinstance(vt.patch_embed, PatchEmbed) or not vt.patch_embed.flatten:
        return "patch embedding must be this package's PatchEmbed(flatten=True)"
    if vt.pos_embed is None:
        return "a learnable position embedding is required"
    if vt.embed_dim % 8 != 0:
        return "embed_dim must be a multiple of 8 for the activation-keeping forward"
    for blk in vt.blocks:
        if blk.attn.head_dim not in (32, 64, 96, 128):
            return f"the attention backward handles head_dim 32, 64, 96 and 128, not {blk.attn.head_dim}"
        if type(blk) is ParallelBlock:
            continue
        if type(blk) is not Block:
            return f"block type {type(blk).__name__} has no backward"
        if blk.mlp.use_conv:
            return "MLP(use_conv=True) has no backward"
        if _ops.activation_kind(blk.mlp.ac1) != _ops._lib.ACT_GELU_ERF:
            return "the backward implements the exact GELU only"
    return None


def _qk_norms(attn, what: str):
    """(q_norm, k_norm) as _ln tuples, or None when the attention has no q/k LayerNorm."""
    qn, kn = _ln(attn.q_norm, what + ".q_norm"), _ln(attn.k_norm, what + ".k_norm")
    if (qn is None) != (kn is None):
        raise NotImplementedError("q_norm and k_norm must both be LayerNorm or both Identity")
    return (qn, kn) if qn is not None else None


def _attention_forward(attn, qkv_pre: torch.Tensor, o: torch.Tensor, B: int, N: int, what: str, s: dict) -> None:
    """Fused attention on [M, 3D] q|k|v rows with the optional per-head q/k LayerNorm; keeps what the backward needs
    (the q|k|v the kernel saw, the pre-norm copy when there is a norm, the log-sum-exp)."""
    H, hd = attn.num_heads, attn.head_dim
    qk = _qk_norms(attn, what)
    qkv = qkv_pre
    if qk is not None:
        qkv = qkv_pre.clone()
        _ops.qk_norm(qkv, B * N, H, hd, qk[0][0], qk[0][1], qk[0][2], qk[1][0], qk[1][1], qk[1][2])
        s["qkv_pre"] = qkv_pre
    lse = torch.empty(B * H * N, dtype=F32, device=qkv.device)
    T.attention_lse(qkv, o, lse, B, N, H, hd, hd ** -0.5)
    s.update(qkv=qkv, o=o, lse=lse)


def _attention_backward(G, attn, s: dict, d_o: torch.Tensor, dqkv: torch.Tensor, B: int, N: int, what: str) -> None:
    """dqkv (an [M, >= 3D] view) <- gradient wrt the q|k|v rows the projection produced."""
    H, hd = attn.num_heads, attn.head_dim
    T.attention_bwd(s["qkv"], s["o"], d_o, s["lse"], dqkv, B, N, H, hd, hd ** -0.5)
    qk = _qk_norms(attn, what)
    if qk is not None:
        T.qk_norm_bwd(s["qkv_pre"], dqkv, B * N, H, hd, qk[0][0], qk[0][2], qk[1][0], qk[1][2],
                      G.buf(qk[0][3]), G.buf(qk[0][4]), G.buf(qk[1][3]), G.buf(qk[1][4]))


# ----------------------------------------------------------------------------------------------------------------
# forward that keeps what the backward needs
# ----------------------------------------------------------------------------------------------------------------
def _block_forward(blk: Block, x: torch.Tensor, B: int, N: int) -> (torch.Tensor, dict):
    D, M, dev, cd = blk.dim, B * N, x.device, x.dtype
    attn, mlp = blk.attn, blk.mlp
    attn._check()
    _check_gelu(mlp.ac1, "MLP")
    H, hd, hid = attn.num_heads, attn.head_dim, mlp.hidden_features
    ln1, ln2 = _ln(blk.norm1, "Block.norm1"), _ln(blk.norm2, "Block.norm2")
    s = {"x": x}
    n1 = x
    if ln1 is not None:
        n1 = _buf(M, D, dev, cd)
        _ops.layernorm(x, n1, M, D, ln1[0], ln1[1], ln1[2])
    qkv = torch.empty((M, 3 * D), dtype=cd, device=dev)
    _ops.gemm(n1, _pk(attn.qkv.weight, attn.qkv.bias, cd), qkv, _ops.EPI_BIAS, M)
    o = _buf(M, D, dev, cd)
    _attention_forward(attn, qkv, o, B, N, "Attention", s)
    # bf16: bias + LayerScale + residual in the projection's epilogue; the branch output z is not kept, the backward gets
    # the LayerScale gradient out of the projection's weight gradient (T.layerscale_grads).  fp32: z kept, separate pass.
    fused_ls = cd == torch.bfloat16
    x1 = _buf(M, D, dev, cd)
    z1 = z2 = None
    if fused_ls:
        _ops.gemm(o, _ops.pack_linear(attn.proj.weight, attn.proj.bias, out_scale=blk.ls1.gamma, dtype=cd), x1,
                  _ops.EPI_RESIDUAL, M, resid=x)
    else:
        z1 = torch.empty((M, _ops.round_up(D, 8)), dtype=cd, device=dev)
        _ops.gemm(o, _pk(attn.proj.weight, attn.proj.bias, cd), z1, _ops.EPI_BIAS, M)
        T.residual_scale(x, z1, _ops.f32vec(blk.ls1.gamma), x1, M, D)
    n2 = x1
    if ln2 is not None:
        n2 = _buf(M, D, dev, cd)
        _ops.layernorm(x1, n2, M, D, ln2[0], ln2[1], ln2[2])
    pre = _buf(M, hid, dev, cd)
    p1 = _pk(mlp.linear_layer1.weight, mlp.linear_layer1.bias, cd)
    h = _buf(M, hid, dev, cd)
    if cd == torch.bfloat16:      # one epilogue writes the pre-activation (kept for the backward) and its GELU
        _ops.gemm(n2, p1, pre, _ops.EPI_BIAS_GELU2, M, resid=h)
    else:
        _ops.gemm(n2, p1, pre, _ops.EPI_BIAS, M)
        _ops.gelu(pre, h, M, _ops.round_up(hid, 8))
    x2 = _buf(M, D, dev, cd)
    if fused_ls:
        _ops.gemm(h, _ops.pack_linear(mlp.linear_layer2.weight, mlp.linear_layer2.bias, out_scale=blk.ls2.gamma, dtype=cd),
                  x2, _ops.EPI_RESIDUAL, M, resid=x1)
    else:
        z2 = torch.empty((M, _ops.round_up(D, 8)), dtype=cd, device=dev)
        _ops.gemm(h, _pk(mlp.linear_layer2.weight, mlp.linear_layer2.bias, cd), z2, _ops.EPI_BIAS, M)
        T.residual_scale(x1, z2, _ops.f32vec(blk.ls2.gamma), x2, M, D)
    s.update(n1=n1, n2=n2, z1=z1, x1=x1, pre=pre, h=h, z2=z2)   # HBM is plentiful: nothing is recomputed in the backward
    return x2, s


def forward_train(model, img: torch.Tensor, compute_dtype: Optional[torch.dtype] = None):
    """(fp32 logits [B, C], saved state) -- the forward of VITC with the activations of the backward kept, in
    ``compute_dtype`` (default: the images' dtype)."""
    vt: VisionTransformer = model.vit_model
    pe: PatchEmbed = vt.patch_embed
    dev = img.device
    B, Cin, Himg, Wimg = img.shape
    assert pe.image_size[0] == Himg, "Input tensor's height does not match the model"
    assert pe.image_size[1] == Wimg, "Input tensor's width does not match the model"
    D, Np = vt.embed_dim, pe.num_patches
    has_cls = vt.class_token is not None
    N = Np + (1 if has_cls else 0)
    M = B * N
    if D % 8 != 0:
        raise NotImplementedError("the training path needs embed_dim % 8 == 0")
    f = _ops.f32vec
    cd = _ops.compute_dtype(img) if compute_dtype is None else compute_dtype
    S = {"B": B, "N": N, "M": M, "D": D, "has_cls": has_cls, "cd": cd}

    pk_pe = _pk(pe.projection.weight, pe.projection.bias, cd)
    A = torch.empty((M, pk_pe.kpad), dtype=cd, device=dev)
    _ops.im2col(img.contiguous(), A, pe.patch_size, pk_pe.kpad, has_cls)       # class-token rows are zero rows
    x = _buf(M, D, dev, cd)
    _ops.gemm(A, pk_pe, x, _ops.EPI_BIAS, M)
    S["A"] = A
    en = _ln(pe.norm_layer, "PatchEmbed.norm_layer")
    npre = _ln(vt.norm_pre, "VisionTransformer.norm_pre")
    S["tok"] = x.clone() if en is not None else None
    cls = f(vt.class_token.reshape(-1)) if has_cls else None
    if en is not None or has_cls:
        _ops.embed_finalize(x, B, N, D, cls, None, en[0] if en else None, en[1] if en else None, en[2] if en else 0.0,
                            None, None, 0.0)
    S["xcat"] = x.clone() if npre is not None else None
    _ops.embed_finalize(x, B, N, D, None, f(vt.pos_embed.reshape(N, D)), None, None, 0.0,
                        npre[0] if npre else None, npre[1] if npre else None, npre[2] if npre else 0.0)
    S["blocks"] = []
    for blk in vt.blocks:
        x, s = (_pblock_forward if type(blk) is ParallelBlock else _block_forward)(blk, x, B, N)
        S["blocks"].append(s)
    S["xL"] = x
    ld = x.shape[1]
    xc = x.view(B, N, ld)[:, 0]                           # class-token rows, row stride N * ld
    fin = _ln(vt.norm_final, "VisionTransformer.norm_final")
    nf = xc
    if fin is not None:
        nf = _buf(B, D, dev, cd)
        _ops.layernorm(xc, nf, B, D, fin[0], fin[1], fin[2])
    S["nf"] = nf
    pk_h = _pk(model.head.weight, model.head.bias, cd)
    logits = torch.empty((B, pk_h.npad), dtype=F32, device=dev)
    _ops.gemm(nf, pk_h, logits, _ops.EPI_BIAS, B)
    return logits[:, :model.head.out_features], S


# ----------------------------------------------------------------------------------------------------------------
# backward
# ----------------------------------------------------------------------------------------------------------------
class _Grads(dict):
    """parameter -> gradient tensor.  With a ``sink`` (parameter -> preallocated fp32 tensor of the parameter's shape,
    e.g. the views of a flat gradient buffer) the kernels write the gradients straight into it."""

    def __init__(self, sink=None):
        super().__init__()
        self.sink = sink

    def buf(self, p: torch.Tensor, *shape) -> torch.Tensor:
        """The (uninitialised) gradient tensor of p, viewed as ``shape`` (default: flat)."""
        t = self.sink.get(p) if self.sink is not None else None
        if t is None:   # no sink, or a frozen parameter (requires_grad False) that the flat buffer does not hold
            t = torch.empty(p.shape, dtype=F32, device=p.device)
        self[p] = t
        return t.view(*shape) if shape else t.view(-1)


def _linear_backward(G: _Grads, lin: nn.Linear, dy: torch.Tensor, x_in: torch.Tensor, M: int, need_dx: bool = True,
                     dy_scale: Optional[torch.Tensor] = None, gelu_input: Optional[torch.Tensor] = None,
                     dx_out: Optional[torch.Tensor] = None):
    """Gradients of y = x W^T + b: G[weight], G[bias]; returns dx [M, >= in_features] (or None).  ``dy_scale`` stands
    for dy * dy_scale (a LayerScale behind the layer); ``gelu_input`` multiplies dx by GELU'(gelu_input)."""
    N, K = lin.weight.shape
    T.linear_dw(dy, x_in, M, N, K, out=G.buf(lin.weight, N, K), bias_out=G.buf(lin.bias) if lin.bias is not None else None,
                dy_scale=dy_scale)
    return T.linear_dx(dy, lin.weight, M, out=dx_out, gelu_input=gelu_input, dy_scale=dy_scale) if need_dx else None


def _scaled_linear_backward(G: _Grads, lin: nn.Linear, ls_gamma: nn.Parameter, dy: torch.Tensor, x_in: torch.Tensor, M: int,
                            gelu_input: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 backward of ``y = resid + gamma * lin(x_in)`` given dy at y, without the Linear's output: the weight-gradient
    GEMM runs without gamma, vitm_layerscale_grads turns its result into G[gamma] and the true G[weight] / G[bias], and
    the input-gradient GEMM folds gamma into W^T.  Returns dx_in."""
    N, K = lin.weight.shape
    dw = G.buf(lin.weight, N, K)
    db = G.buf(lin.bias) if lin.bias is not None else None
    T.linear_dw(dy, x_in, M, N, K, out=dw, bias_out=db)
    gam = _ops.f32vec(ls_gamma)
    T.layerscale_grads(lin.weight, lin.bias, dw, db, gam, G.buf(ls_gamma), round_bf16=True)
    return T.linear_dx(dy, lin.weight, M, gelu_input=gelu_input, dy_scale=gam)


def _scaled(dx: torch.Tensor, gamma: torch.Tensor, M: int, D: int):
    """(dy, dy_scale) for the Linear in front of a LayerScale, given the gradient behind it: bf16 keeps dx and hands the
    scale to the GEMMs (folded into W^T and into the rows of dW); fp32 materialises dx * gamma."""
    if dx.dtype == torch.bfloat16:
        return dx, gamma
    D8 = _ops.round_up(D, 8)
    dz = torch.empty((M, D8), dtype=dx.dtype, device=dx.device)
    _ops.scale_channels(dx[:, :D8] if dx.shape[1] != D8 else dx, dz, M, D8, _pad8(gamma))
    return dz, None


def _block_backward(G: _Grads, blk: Block, s: dict, dx2: torch.Tensor, B: int, N: int) -> torch.Tensor:
    D, M, dev, cd = blk.dim, B * N, dx2.device, dx2.dtype
    attn, mlp = blk.attn, blk.mlp
    H, hd, hid = attn.num_heads, attn.head_dim, mlp.hidden_features
    ln1, ln2 = _ln(blk.norm1, "Block.norm1"), _ln(blk.norm2, "Block.norm2")
    g1, g2 = _ops.f32vec(blk.ls1.gamma), _ops.f32vec(blk.ls2.gamma)

    # ---- MLP branch: x2 = x1 + g2 * z2, z2 = fc2(gelu(pre)), pre = fc1(LN2(x1))
    if cd == torch.bfloat16:      # GELU backward in the epilogue of the input-gradient GEMM; LayerScale from the dW
        dpre = _scaled_linear_backward(G, mlp.linear_layer2, blk.ls2.gamma, dx2, s["h"], M, gelu_input=s["pre"])
    else:
        T.colsum(dx2, M, D, s["z2"], out=G.buf(blk.ls2.gamma))
        dz2, sc2 = _scaled(dx2, g2, M, D)
        dh = _linear_backward(G, mlp.linear_layer2, dz2, s["h"], M)
        dpre = dh                                         # in place: element-wise
        T.gelu_bwd(s["pre"], dh, dpre, M, hid)
    dn2 = _linear_backward(G, mlp.linear_layer1, dpre, s["n2"], M)
    dx1 = _buf(M, D, dev, cd)
    if ln2 is not None:
        T.layernorm_bwd(s["x1"], dn2, ln2[0], ln2[2], M, D, dx1, dx_add=dx2, dw_out=G.buf(ln2[3]), db_out=G.buf(ln2[4]))
    else:
        _add(dx2, dn2, dx1, M, D)

    # ---- attention branch: x1 = x + g1 * z1, z1 = proj(attn(qkv)), qkv = Wqkv LN1(x)
    if cd == torch.bfloat16:
        d_o = _scaled_linear_backward(G, attn.proj, blk.ls1.gamma, dx1, s["o"], M)
    else:
        T.colsum(dx1, M, D, s["z1"], out=G.buf(blk.ls1.gamma))
        dz1, sc1 = _scaled(dx1, g1, M, D)
        d_o = _linear_backward(G, attn.proj, dz1, s["o"], M, dy_scale=sc1)
    dqkv = torch.empty((M, 3 * D), dtype=cd, device=dev)
    _attention_backward(G, attn, s, d_o, dqkv, B, N, "Attention")
    dn1 = _linear_backward(G, attn.qkv, dqkv, s["n1"], M)
    dx = _buf(M, D, dev, cd)
    if ln1 is not None:
        T.layernorm_bwd(s["x"], dn1, ln1[0], ln1[2], M, D, dx, dx_add=dx1, dw_out=G.buf(ln1[3]), db_out=G.buf(ln1[4]))
    else:
        _add(dx1, dn1, dx, M, D)
    return dx


# ---- ViT-22B style parallel block (parallel_block.py:193-221) --------------------------------------------------------
def _pblock_forward(blk: ParallelBlock, x: torch.Tensor, B: int, N: int) -> (torch.Tensor, dict):
    D, M, dev, cd = blk.dim, B * N, x.device, x.dtype
    blk.attn._check()
    _check_gelu(blk.mlp.act_layer, "ParallelBlock.mlp")
    if D % 8 != 0:
        raise NotImplementedError("ParallelBlock on the HIP path needs dim % 8 == 0")
    hid = blk.mlp_hidden_dim
    ln = _ln(blk.norm, "ParallelBlock.norm")
    s = {"x": x}
    n = x
    if ln is not None:
        n = _buf(M, D, dev, cd)
        _ops.layernorm(x, n, M, D, ln[0], ln[1], ln[2])
    w1, b1 = blk.fused_proj_1.weight, blk.fused_proj_1.bias
    qkv = torch.empty((M, 3 * D), dtype=cd, device=dev)
    _ops.gemm(n, _pk(w1[:3 * D], b1[:3 * D] if b1 is not None else None, cd), qkv, _ops.EPI_BIAS, M)
    pre = _buf(M, hid, dev, cd)
    h = _buf(M, hid, dev, cd)
    pk_mlp = _pk(w1[3 * D:], b1[3 * D:] if b1 is not None else None, cd)
    if cd == torch.bfloat16:
        _ops.gemm(n, pk_mlp, pre, _ops.EPI_BIAS_GELU2, M, resid=h)
    else:
        _ops.gemm(n, pk_mlp, pre, _ops.EPI_BIAS, M)
        _ops.gelu(pre, h, M, _ops.round_up(hid, 8))
    o = _buf(M, D, dev, cd)
    _attention_forward(blk.attn, qkv, o, B, N, "ParallelBlockMultiHeadAttention", s)
    cat = _buf(M, 2 * D, dev, cd)                            # attention half | MLP half: the concat is free
    _ops.gemm(o, _pk(blk.attn.proj.weight, blk.attn.proj.bias, cd), cat[:, :D], _ops.EPI_BIAS, M)
    _ops.gemm(h, _pk(blk.mlp.out_proj.weight, blk.mlp.out_proj.bias, cd), cat[:, D:2 * D], _ops.EPI_BIAS, M)
    x2 = _buf(M, D, dev, cd)
    z = None
    if cd == torch.bfloat16:      # bias + LayerScale + residual in the epilogue, z not kept (see _block_forward)
        _ops.gemm(cat, _ops.pack_linear(blk.fused_proj_2.weight, blk.fused_proj_2.bias, out_scale=blk.layer_scale.gamma, dtype=cd),
                  x2, _ops.EPI_RESIDUAL, M, resid=x)
    else:
        z = torch.empty((M, D), dtype=cd, device=dev)
        _ops.gemm(cat, _pk(blk.fused_proj_2.weight, blk.fused_proj_2.bias, cd), z, _ops.EPI_BIAS, M)
        T.residual_scale(x, z, _ops.f32vec(blk.layer_scale.gamma), x2, M, D)
    s.update(n=n, pre=pre, h=h, cat=cat, z=z)
    return x2, s


def _pblock_backward(G: _Grads, blk: ParallelBlock, s: dict, dx2: torch.Tensor, B: int, N: int) -> torch.Tensor:
    D, M, dev, cd = blk.dim, B * N, dx2.device, dx2.dtype
    hid = blk.mlp_hidden_dim
    ln = _ln(blk.norm, "ParallelBlock.norm")
    gam = _ops.f32vec(blk.layer_scale.gamma)
    # x2 = x + gamma * z, z = fused_proj_2(cat)
    if cd == torch.bfloat16:
        dcat = _scaled_linear_backward(G, blk.fused_proj_2, blk.layer_scale.gamma, dx2, s["cat"], M)    # [M, >= 2D]
    else:
        T.colsum(dx2, M, D, s["z"], out=G.buf(blk.layer_scale.gamma))
        dz, sc = _scaled(dx2, gam, M, D)
        dcat = _linear_backward(G, blk.fused_proj_2, dz, s["cat"], M, dy_scale=sc)    # [M, >= 2D]
    # attention half: cat[:, :D] = attn.proj(o)
    d_o = _linear_backward(G, blk.attn.proj, dcat[:, :D], s["o"], M)
    # MLP half: cat[:, D:2D] = out_proj(gelu(pre))
    # gradient of the fused first projection's output, [q | k | v | hidden] side by side like its weight rows
    d1 = _buf(M, 3 * D + hid, dev, cd)
    if cd == torch.bfloat16:      # GELU backward in the epilogue of the input-gradient GEMM, written in place
        _linear_backward(G, blk.mlp.out_proj, dcat[:, D:2 * D], s["h"], M, gelu_input=s["pre"],
                         dx_out=d1[:, 3 * D:3 * D + _ops.round_up(hid, 8)])
    else:
        dh = _linear_backward(G, blk.mlp.out_proj, dcat[:, D:2 * D], s["h"], M)
        T.gelu_bwd(s["pre"], dh, d1[:, 3 * D:3 * D + hid], M, hid)
    _attention_backward(G, blk.attn, s, d_o, d1[:, :3 * D], B, N, "ParallelBlockMultiHeadAttention")
    dn = _linear_backward(G, blk.fused_proj_1, d1, s["n"], M)
    dx = _buf(M, D, dev, cd)
    if ln is not None:
        T.layernorm_bwd(s["x"], dn, ln[0], ln[2], M, D, dx, dx_add=dx2, dw_out=G.buf(ln[3]), db_out=G.buf(ln[4]))
    else:
        _add(dx2, dn, dx, M, D)
    return dx


def _pad8(v: torch.Tensor) -> torch.Tensor:
    n = v.numel()
    n8 = _ops.round_up(n, 8)
    if n8 == n:
        return v
    out = torch.zeros(n8, dtype=F32, device=v.device)
    out[:n] = v
    return out


def _add(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, D: int) -> None:
    """out = a + b over [M, D] (residual_scale with a unit LayerScale)."""
    if b.shape[1] > out.shape[1]:
        b = b[:, :out.shape[1]]
    ones = torch.ones(D, dtype=F32, device=a.device)
    T.residual_scale(a, b, ones, out, M, D)


def backward_train(model, S: dict, dlogits: torch.Tensor, sink=None, on_ready=None) -> Dict[nn.Parameter, torch.Tensor]:
    """Gradients of every parameter given d(loss)/d(logits) [B, C]; written into ``sink`` (parameter -> fp32 tensor)
    when one is given, into fresh tensors otherwise.  ``on_ready(params)`` is called as soon as the gradients of a
    group of parameters (head + final norm, then each block from the last to the first) are queued."""
    vt: VisionTransformer = model.vit_model
    pe: PatchEmbed = vt.patch_embed
    B, N, M, D, has_cls, cd = S["B"], S["N"], S["M"], S["D"], S["has_cls"], S["cd"]
    dev = dlogits.device
    G = _Grads(sink)
    dl = dlogits.contiguous()
    if cd != F32:
        dl = _ops.to_bf16(dl)

    # ---- head and final norm (class-token rows only)
    dnf = _linear_backward(G, model.head, dl, S["nf"], B)
    xL = S["xL"]
    ld = xL.shape[1]
    dx = torch.zeros((M, ld), dtype=cd, device=dev)
    dxc = dx.view(B, N, ld)[:, 0]                          # gradient lands in the class-token rows
    fin = _ln(vt.norm_final, "VisionTransformer.norm_final")
    if fin is not None:
        T.layernorm_bwd(xL.view(B, N, ld)[:, 0], dnf, fin[0], fin[2], B, D, dxc, dw_out=G.buf(fin[3]), db_out=G.buf(fin[4]))
    else:
        dxc.copy_(dnf[:, :ld] if dnf.shape[1] >= ld else torch.nn.functional.pad(dnf, (0, ld - dnf.shape[1])))

    if on_ready is not None:
        on_ready(list(model.head.parameters()) + list(vt.norm_final.parameters()))

    # ---- blocks, last to first
    for blk, s in zip(reversed(list(vt.blocks)), reversed(S["blocks"])):
        dx = (_pblock_backward if type(blk) is ParallelBlock else _block_backward)(G, blk, s, dx, B, N)
        if on_ready is not None:
            on_ready(list(blk.parameters()))

    # ---- x0 = LN_pre(xcat) + pos ; xcat = [cls ; LN_e(tok)] ; tok = im2col(img) Wpe^T (+ b)
    dpos = T.colsum(dx.view(B, N * ld), B, N * ld)         # sum over the batch
    G.buf(vt.pos_embed, N, D).copy_(dpos.view(N, ld)[:, :D])
    npre = _ln(vt.norm_pre, "VisionTransformer.norm_pre")
    dxcat = dx
    if npre is not None:
        dxcat = _buf(M, D, dev, cd)
        T.layernorm_bwd(S["xcat"], dx, npre[0], npre[2], M, D, dxcat, dw_out=G.buf(npre[3]), db_out=G.buf(npre[4]))
    if has_cls:
        T.colsum(dxcat.view(B, N * ld), B, D, out=G.buf(vt.class_token))
        dxcat.view(B, N, ld)[:, 0].zero_()                 # the class-token rows carry no patch gradient
    en = _ln(pe.norm_layer, "PatchEmbed.norm_layer")
    dtok = dxcat
    if en is not None:
        dtok = _buf(M, D, dev, cd)
        T.layernorm_bwd(S["tok"], dxcat, en[0], en[2], M, D, dtok, dw_out=G.buf(en[3]), db_out=G.buf(en[4]))
        if has_cls:
            dtok.view(B, N, ld)[:, 0].zero_()
    Wpe = pe.projection.weight
    Dp, Kp = Wpe.shape[0], Wpe[0].numel()
    T.linear_dw(dtok, S["A"], M, Dp, Kp, out=G.buf(Wpe, Dp, Kp),
                bias_out=G.buf(pe.projection.bias) if pe.projection.bias is not None else None)
    return G


# ----------------------------------------------------------------------------------------------------------------
# autograd hook-up
# ----------------------------------------------------------------------------------------------------------------
class _VitcFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, img, compute_dtype, *params):
        with torch.cuda.device(img.device):
            logits, S = forward_train(model, img, compute_dtype)
        ctx.model, ctx.S, ctx.params = model, S, params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        with torch.cuda.device(dlogits.device):
            G = backward_train(ctx.model, ctx.S, dlogits)
        ctx.S = None
        return (None, None, None) + tuple(G.get(p).reshape(p.shape) if G.get(p) is not None else None for p in ctx.params)


def _check_trainable(model, img: torch.Tensor, compute_dtype: Optional[torch.dtype]):
    """(compute dtype, parameter list) or an exception that says why this model / input cannot take the training path."""
    why = supported(model)
    if why is not None:
        raise NotImplementedError(f"training path: {why}")
    _ops.require_cuda(img, "training step")
    cd = _ops.compute_dtype(img) if compute_dtype is None else compute_dtype
    if cd not in (F32, torch.bfloat16):
        raise TypeError(f"compute_dtype {cd} unsupported (torch.float32 or torch.bfloat16)")
    params = [p for p in model.parameters()]
    if any(p.dtype != F32 for p in params):
        raise TypeError("the training path needs fp32 (master) parameters, also for bf16 compute")
    return cd, params


def vitc_forward_with_grad(model, img: torch.Tensor, compute_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``model(img)`` with an autograd graph: fp32 logits whose ``backward`` fills every parameter's ``.grad``.
    ``compute_dtype`` (default: the images' dtype) picks fp32 or bf16 mixed precision; parameters are fp32 in both."""
    cd, params = _check_trainable(model, img, compute_dtype)
    return _VitcFunction.apply(model, img, cd, *params)


class _CrossEntropyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        with torch.cuda.device(logits.device):
            loss, dl = T.cross_entropy(logits.contiguous(), labels, want_grad=True)
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None      # g is the scalar 1 of loss.backward(); a torch multiply of a [B, C] tensor by it


class CrossEntropyLoss(nn.Module):
    """``nn.CrossEntropyLoss()`` with class-index targets and mean reduction (train.py:187), as a HIP kernel."""

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        _ops.require_cuda(logits, "CrossEntropyLoss.forward")
        if logits.dtype != F32 or logits.dim() != 2:
            raise TypeError("CrossEntropyLoss expects fp32 logits of shape (B, C)")
        return _CrossEntropyFunction.apply(logits, labels)


# ----------------------------------------------------------------------------------------------------------------
# gradients in one flat buffer, clipping, AdamW, data-parallel all-reduce
# ----------------------------------------------------------------------------------------------------------------
FLAT_ALIGN = 64     # floats: every tensor starts on a 256-byte boundary of the flat buffer (kernels use 16-byte accesses)


def _flat_layout(params):
    """(offsets, total) in floats of the tensors laid end to end, each start rounded up to FLAT_ALIGN; the gaps stay
    zero for ever (zero gradient, zero moments, zero parameter: AdamW maps that to zero)."""
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
    return offsets, off


class FlatGradients:
    """Every parameter's ``.grad`` as a view into one flat fp32 buffer: one kernel clips them all, one collective
    all-reduces them all.  ``zero()`` replaces ``optimizer.zero_grad()``."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradients: no parameter requires a gradient")
        self.offsets, total = _flat_layout(self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=F32, device=dev)
        self.views = {}
        for p, off in zip(self.params, self.offsets):
            p.grad = self.views[p] = self.flat[off:off + p.numel()].view(p.shape)

    def attach(self) -> None:
        """Make every ``.grad`` the view into the flat buffer again (``zero_grad(set_to_none=True)`` drops them)."""
        for p in self.params:
            if p.grad is not self.views[p]:
                p.grad = self.views[p]

    def zero(self) -> None:
        """Replaces ``optimizer.zero_grad()``."""
        self.flat.zero_()
        self.attach()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ (train.py:222): returns the total norm (a device scalar)."""
        with torch.cuda.device(self.flat.device):
            return T.grad_clip(self.flat, max_norm)[0]

    def all_reduce(self, group=None) -> None:
        """Average the gradients over the data-parallel group (RCCL over xGMI when the backend is nccl)."""
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
        else:                                   # gloo rehearsal on CPU tensors: sum, then divide
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


class GradientReducer:
    """Data-parallel averaging of a FlatGradients buffer, overlapped with the backward pass.

    The backward produces gradients from the last parameters of the model to the first, and parameters lie in the flat
    buffer in registration order, so what is ready at any time is a suffix of what is still missing: ``ready(params)``
    extends the pending range downwards and, once it holds ``bucket_bytes``, starts an asynchronous all-reduce of that
    contiguous slice (no copies, no per-tensor collectives) while the backward kernels of earlier layers keep running.
    ``finish()`` reduces what is left and waits.  RCCL over xGMI when the backend is nccl (one process per GPU); over
    point-to-point xGMI links a collective is per-link bound, so buckets are tens of MB, not the 1 MB of NVSwitch
    habits."""

    def __init__(self, grads: "FlatGradients", group=None, bucket_bytes: int = 64 << 20, single_rank: bool = False):
        """``single_rank``: run the collectives even in a group of one rank (an average over one replica changes
        nothing; it lets one GPU rehearse the bucketing, the stream ordering and the RCCL calls)."""
        import torch.distributed as dist
        self.grads, self.group, self.bucket = grads, group, int(bucket_bytes)
        self.active = (dist.is_available() and dist.is_initialized()
                       and (dist.get_world_size(group) > 1 or bool(single_rank)))
        self.world = dist.get_world_size(group) if self.active else 1
        self.avg = self.active and dist.get_backend(group) == "nccl"
        self.offset = {p: o for p, o in zip(grads.params, grads.offsets)}
        self.begin()

    def begin(self) -> None:
        self.hi = self.lo = self.grads.flat.numel()     # pending range [lo, hi): ready but not yet sent
        self.sent_lo = self.hi                          # everything in [sent_lo, end) has been handed to a collective
        self.works = []

    def _fire(self, lo: int, hi: int) -> None:
        import torch.distributed as dist
        if hi <= lo:
            return
        seg = self.grads.flat[lo:hi]
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.works.append((dist.all_reduce(seg, op=op, group=self.group, async_op=True), seg))
        self.sent_lo = lo

    def ready(self, params) -> None:
        """The gradients of ``params`` are final (their kernels are queued on the current stream)."""
        if not self.active:
            return
        lo = min((self.offset[p] for p in params if p in self.offset), default=self.lo)
        if lo < self.lo:
            self.lo = lo
        if (self.hi - self.lo) * 4 >= self.bucket:
            self._fire(self.lo, self.hi)
            self.hi = self.lo

    def finish(self) -> None:
        """Reduce whatever has not been sent (down to the start of the buffer) and wait for every collective."""
        if not self.active:
            return
        self._fire(0, self.hi)
        for work, seg in self.works:
            work.wait()
            if not self.avg:                    # gloo rehearsal: sum, then divide
                seg.div_(self.world)
        self.begin()


class FlatParameters:
    """Every parameter's storage moved (values kept) into one flat fp32 buffer, each ``p.data`` a view of it: one kernel
    can then update the whole model.  ``load_state_dict`` and in-place edits keep the views; ``model.to()`` / ``.float()``
    re-allocate the parameters and so detach them (the optimizer notices and falls back to one launch per tensor)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatParameters: no parameter requires a gradient")
        if any(p.dtype != F32 or not p.is_cuda for p in self.params):
            raise TypeError("FlatParameters holds fp32 CUDA parameters")
        self.offsets, total = _flat_layout(self.params)
        self.flat = torch.zeros(total, dtype=F32, device=self.params[0].device)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                v = self.flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v

    def intact(self) -> bool:
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (train.py:68-72) with the update in a HIP kernel; state_dict layout is torch's, so optimizer
    states saved by the reference load and the torch LR schedulers drive it."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = None

    # ---- one launch for the whole model when parameters and gradients each live in one flat buffer ---------------
    def use_flat(self, params: "FlatParameters", grads: "FlatGradients") -> bool:
        """Update all parameters with one kernel launch from now on.  Needs a single parameter group holding exactly
        the tensors of ``params`` / ``grads`` in their order; returns False (and changes nothing) otherwise."""
        if len(self.param_groups) != 1:
            return False
        ps = self.param_groups[0]["params"]
        if len(ps) != len(params.params) or any(a is not b for a, b in zip(ps, params.params)) \
                or len(ps) != len(grads.params) or any(a is not b for a, b in zip(ps, grads.params)) \
                or params.offsets != grads.offsets:
            return False
        m, v = torch.zeros_like(params.flat), torch.zeros_like(params.flat)
        self._flat = (params, grads, m, v)
        self._seed_flat_state()
        return True

    def _seed_flat_state(self) -> None:
        """Per-parameter state entries become views of the flat moment buffers (torch's state_dict layout is kept);
        existing state (a loaded checkpoint) is copied in first.  Every parameter keeps a step counter of its own (as
        torch.optim.AdamW does: a state_dict saved here must load into it), all advanced together by ``_flat_step``."""
        params, _, m, v = self._flat
        step = None
        for p, off in zip(params.params, params.offsets):
            st = self.state[p]
            mv, vv = m[off:off + p.numel()].view(p.shape), v[off:off + p.numel()].view(p.shape)
            if len(st) != 0:
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
                step = st["step"] if step is None else step
            st["exp_avg"], st["exp_avg_sq"] = mv, vv
        step = 0.0 if step is None else float(step)
        for p in params.params:
            self.state[p]["step"] = torch.tensor(step)

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        # a checkpoint whose parameters share ONE step tensor (written by an older flat-mode optimizer) must not stay
        # aliased: every in-place "step += 1" would then count once per parameter
        seen = set()
        for st in self.state.values():
            t = st.get("step")
            if torch.is_tensor(t):
                if id(t) in seen:
                    st["step"] = t.clone()
                seen.add(id(st["step"]))
        if self._flat is not None:
            self._seed_flat_state()

    def _flat_ok(self) -> bool:
        params, grads, _, _ = self._flat
        return params.intact() and all(p.grad is not None and p.grad.data_ptr() == grads.views[p].data_ptr() for p in params.params)

    def _flat_advance(self) -> int:
        """Host side of one flat update: every parameter's step counter moves on, cached packed weights are invalidated.
        Returns the new step number."""
        params = self._flat[0]
        for p in params.params:
            self.state[p]["step"] += 1
            torch.autograd.graph.increment_version(p)       # cached packed weights must be rebuilt
        return int(self.state[params.params[0]]["step"].item())

    def _flat_step(self) -> bool:
        if not self._flat_ok():
            return False
        params, grads, m, v = self._flat
        group = self.param_groups[0]
        step = self._flat_advance()
        b1, b2 = group["betas"]
        with torch.cuda.device(params.flat.device):
            T.adamw(params.flat, grads.flat, m, v, group["lr"], b1, b2, group["eps"], group["weight_decay"], step)
        return True

    def flat_launch_from(self, hyper: torch.Tensor) -> None:
        """Only the device side of a flat update, with its scalars read from ``hyper`` (7 fp32 on the device, see
        ``next_hyper``): the launch a captured training step contains."""
        params, grads, m, v = self._flat
        with torch.cuda.device(params.flat.device):
            T.adamw_hyper(params.flat, grads.flat, m, v, hyper)

    def next_hyper(self):
        """Advance the step counters (host side of a flat update) and return the seven scalars of that update for the
        current learning rate."""
        group = self.param_groups[0]
        step = self._flat_advance()
        b1, b2 = group["betas"]
        return T.adamw_hyper_values(group["lr"], b1, b2, group["eps"], group["weight_decay"], step)

    def _unflatten_state(self) -> None:
        """Leave flat mode for good: the state entries get storage of their own and a step counter each."""
        params = self._flat[0]
        self._flat = None
        for p in params.params:
            st = self.state[p]
            if len(st) != 0:
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
                st["step"] = st["step"].clone()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._flat is not None:
            if self._flat_step():
                return loss
            self._unflatten_state()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != F32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("vitm_amd AdamW updates contiguous fp32 CUDA parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                with torch.cuda.device(p.device):
                    T.adamw(p, p.grad, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                            group["weight_decay"], int(st["step"].item()))
                torch.autograd.graph.increment_version(p)   # the kernel wrote p: cached packed weights must be rebuilt
        return loss


def reference_optimizer_and_scheduler(model, total_iters: int):
    """The optimizer / scheduler of the reference's ``getOptimizerAndSchedular`` (train.py:65-93)."""
    opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    sched = torch.optim.lr_scheduler.SequentialLR(
        opt, [torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=500),
              torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(1, total_iters - 500), eta_min=1e-6)],
        milestones=[500])
    return opt, sched


class Trainer:
    """One object for the loop body of train.py:215-224 (plus the data-parallel all-reduce)."""

    def __init__(self, model, optimizer, scheduler=None, max_norm: float = 1.0, group=None,
                 compute_dtype: Optional[torch.dtype] = None, bucket_bytes: int = 64 << 20, reduce_single_rank: bool = False):
        self.model, self.optimizer, self.scheduler, self.max_norm, self.group = model, optimizer, scheduler, max_norm, group
        self.compute_dtype = compute_dtype          # None: follow the images (fp32 images -> fp32, bf16 -> mixed precision)
        self.grads = FlatGradients(model.parameters())
        self.params = None
        if isinstance(optimizer, AdamW):             # one flat parameter buffer -> one AdamW launch per step
            flat = FlatParameters(model.parameters())
            if optimizer.use_flat(flat, self.grads):
                self.params = flat
        self.criterion = CrossEntropyLoss()
        self.last_grad_norm = None
        self.reducer = GradientReducer(self.grads, group, bucket_bytes, single_rank=reduce_single_rank)
        if self.reducer.active:
            self.sync_from_rank0()

    def sync_from_rank0(self) -> None:
        """Data-parallel replicas must start identical: rank 0's parameters (and floating-point buffers) replace every
        other rank's, as DistributedDataParallel does at construction.  Optimizer state is NOT synchronised: create the
        optimizer fresh on every rank, or load the same checkpoint on every rank."""
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        with torch.no_grad():
            if self.params is not None and self.params.intact():
                dist.broadcast(self.params.flat, src=src, group=self.group)
                frozen = [p for p in self.model.parameters() if not p.requires_grad]
            else:
                frozen = list(self.model.parameters())
            for t in frozen + [b for b in self.model.buffers() if b.is_floating_point()]:
                dist.broadcast(t.data, src=src, group=self.group)
            for p in self.model.parameters():
                torch.autograd.graph.increment_version(p)   # cached packed weights must be rebuilt

    def device_step(self, images: torch.Tensor, labels: torch.Tensor, hyper: torch.Tensor):
        """The kernels of a step and nothing else -- forward, loss, backward, clip, AdamW with its scalars from the device
        tensor ``hyper`` -- for capture into a HIP graph (vitm_amd.graph.GraphedTrainStep does the host side around each
        replay).  Single process, flat AdamW only.  Returns (loss, gradient norm), both device scalars."""
        if self.params is None or self.reducer.active:
            raise RuntimeError("device_step needs the flat AdamW of this package and no gradient all-reduce")
        cd, _ = _check_trainable(self.model, images, self.compute_dtype)
        with torch.no_grad(), torch.cuda.device(images.device):
            self.grads.attach()
            logits, S = forward_train(self.model, images, cd)
            loss, dl = T.cross_entropy(logits, labels, want_grad=True)
            G = backward_train(self.model, S, dl, sink=self.grads.views)
            for p in self.grads.params:
                if p not in G:
                    self.grads.views[p].zero_()
            norm = self.grads.clip_grad_norm_(self.max_norm)
            self.optimizer.flat_launch_from(hyper)
        return loss.reshape(()), norm

    def step(self, images: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """zero_grad, forward, loss, backward, all-reduce, clip, optimizer and scheduler step.  The backward kernels
        write every gradient straight into the flat buffer (no autograd graph, no accumulation pass)."""
        cd, _ = _check_trainable(self.model, images, self.compute_dtype)
        with torch.no_grad(), torch.cuda.device(images.device):
            self.grads.attach()
            logits, S = forward_train(self.model, images, cd)
            loss, dl = T.cross_entropy(logits, labels, want_grad=True)
            G = backward_train(self.model, S, dl, sink=self.grads.views, on_ready=self.reducer.ready)
            loss = loss.reshape(())
            self.reducer.finish()                    # the gradient all-reduce ran in buckets under the backward
            for p in self.grads.params:              # a parameter the backward did not reach keeps no stale gradient
                if p not in G:
                    self.grads.views[p].zero_()
        self.last_grad_norm = self.grads.clip_grad_norm_(self.max_norm)
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss.detach()
