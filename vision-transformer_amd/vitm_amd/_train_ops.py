"""Torch-tensor front end of the training entry points of the C ABI (backward pieces, loss, clip, AdamW).

Activation-sized tensors are fp32 or bf16 (one dtype per call); statistics, reductions, parameter gradients, the loss
and the optimizer state are always fp32."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._ops import _p, _dt, _stream, round_up, k_multiple, pack_linear, gemm, to_f32, to_bf16, EPI_BIAS, EPI_DGELU, PackedLinear

COLSUM_CHUNKS = 64
COLSUM_WORKGROUPS = 1024
LN_BWD_BLOCKS = 1024
F32 = torch.float32


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32 and t.stride(-1) == 1, "fp32 row-major tensor expected"
    return t


def _rm(*ts: torch.Tensor) -> int:
    """dtype code shared by row-major operands of one call."""
    d = ts[0].dtype
    for t in ts:
        assert t.dtype == d and t.stride(-1) == 1, "row-major tensors of one dtype expected"
    return _dt(ts[0])


def attention_lse(qkv, out, lse, B, N, H, hd, scale) -> None:
    lib = _lib.load()
    if qkv.dtype == torch.bfloat16:
        assert out.dtype == torch.bfloat16
        _lib.check(lib.vitm_attention_lse(_p(qkv), qkv.stride(0), _p(out), out.stride(0), _p(_f32(lse)), B, N, H, hd,
                                          float(scale), _stream()), "vitm_attention_lse")
        return
    _lib.check(lib.vitm_attention_f32_lse(_p(_f32(qkv)), qkv.stride(0), _p(_f32(out)), out.stride(0), _p(lse), B, N, H,
                                          hd, float(scale), _stream()), "vitm_attention_f32_lse")


def attention_bwd(qkv, o, d_o, lse, dqkv, B, N, H, hd, scale) -> None:
    lib = _lib.load()
    if qkv.dtype == torch.bfloat16:
        assert o.dtype == d_o.dtype == dqkv.dtype == torch.bfloat16
        _lib.check(lib.vitm_attention_bwd(_p(qkv), qkv.stride(0), _p(o), o.stride(0), _p(d_o), d_o.stride(0), _p(_f32(lse)),
                                          _p(dqkv), dqkv.stride(0), _p(_delta(qkv, B, H, N)), B, N, H, hd, float(scale),
                                          _stream()), "vitm_attention_bwd")
        return
    _lib.check(lib.vitm_attention_bwd_f32(_p(_f32(qkv)), qkv.stride(0), _p(_f32(o)), o.stride(0), _p(_f32(d_o)),
                                          d_o.stride(0), _p(lse), _p(_f32(dqkv)), dqkv.stride(0), _p(_delta(qkv, B, H, N)),
                                          B, N, H, hd, float(scale), _stream()), "vitm_attention_bwd_f32")


def _delta(ref, B, H, N):
    return torch.empty(B * H * N, dtype=F32, device=ref.device)


def transpose(src: torch.Tensor, M: int, N: int, m_pad: Optional[int] = None, out_dtype: Optional[torch.dtype] = None,
              rows: Optional[int] = None, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M, N] (row stride src.stride(0)) -> new [rows or N, m_pad] in out_dtype (default: the source's) with the pad
    columns (and pad rows) zero; ``row_scale`` (M floats) multiplies source row m."""
    m_pad = M if m_pad is None else m_pad
    rows = N if rows is None else rows
    dt = src.dtype if out_dtype is None else out_dtype
    dst = (torch.empty if rows == N else torch.zeros)((rows, m_pad), dtype=dt, device=src.device)
    assert src.stride(-1) == 1
    _lib.check(_lib.load().vitm_transpose(_p(src), _dt(src), src.stride(0), _p(dst), _dt(dst), m_pad, M, N, m_pad,
                                          _p(None if row_scale is None else _f32(row_scale)), _stream()), "vitm_transpose")
    return dst


_ZEROS = {}


def _zeros_f32(device, n: int) -> torch.Tensor:
    """A shared all-zero fp32 vector (the bias of a GEMM that has none); never written."""
    key = (device, n)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(n, dtype=F32, device=device)
    return z


def colsum(a: torch.Tensor, M: int, D: int, b: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[d] = sum_m a[m, d] * (b[m, d] if b is given else 1) in fp32, deterministic two-stage reduction."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(D, dtype=F32, device=a.device)
    dt = _rm(a) if b is None else _rm(a, b)
    # row chunks: enough workgroups (one per 256 columns x chunk) to fill the chip, at least 16 rows each
    nch = min(max(COLSUM_CHUNKS, COLSUM_WORKGROUPS // ((D + 255) // 256)), 512, M // 16) if M >= 32 else 1
    ldb = 0 if b is None else b.stride(0)
    if nch > 1:
        part = torch.empty((nch, D), dtype=F32, device=a.device)
        _lib.check(lib.vitm_colsum(_p(a), a.stride(0), _p(b), ldb, dt, _p(part), D, M, D, nch, _stream()), "vitm_colsum")
        _lib.check(lib.vitm_colsum_pair(_p(part), None, nch, D, _p(_f32(out)), None, _stream()), "vitm_colsum_pair")
    else:
        _lib.check(lib.vitm_colsum(_p(a), a.stride(0), _p(b), ldb, dt, _p(out), D, M, D, 1, _stream()), "vitm_colsum")
    return out


def layernorm_bwd(x, dy, w, eps, M, D, dx, dx_add=None, dw_out=None, db_out=None):
    """dx (= dx_add + LayerNorm backward) and, when dw_out / db_out are given, the parameter gradients."""
    nb = min(LN_BWD_BLOCKS, (M + 3) // 4)
    dwp = torch.empty((nb, D), dtype=F32, device=x.device)
    dbp = torch.empty((nb, D), dtype=F32, device=x.device)
    dt = _rm(x, dy, dx) if dx_add is None else _rm(x, dy, dx, dx_add)
    _lib.check(_lib.load().vitm_layernorm_bwd(
        _p(x), x.stride(0), _p(dy), dy.stride(0), _p(_f32(w)), float(eps), _p(dx_add),
        0 if dx_add is None else dx_add.stride(0), _p(dx), dx.stride(0), dt, _p(dwp), _p(dbp), nb, M, D, _stream()),
        "vitm_layernorm_bwd")
    if dw_out is not None and db_out is not None:
        _lib.check(_lib.load().vitm_colsum_pair(_p(dwp), _p(dbp), nb, D, _p(_f32(dw_out)), _p(_f32(db_out)), _stream()),
                   "vitm_colsum_pair")
        return
    if dw_out is not None:
        colsum(dwp, nb, D, out=dw_out)
    if db_out is not None:
        colsum(dbp, nb, D, out=db_out)


QK_NORM_BWD_BLOCKS = 256


def qk_norm_bwd(qkv_pre, dqkv, M, H, hd, q_w, q_eps, k_w, k_eps, dqw_out, dqb_out, dkw_out, dkb_out) -> None:
    """In place on dqkv: gradients wrt the normalised q, k become gradients wrt the pre-norm q, k; the four parameter
    gradients (head_dim floats each) go to the *_out tensors."""
    nb = min(QK_NORM_BWD_BLOCKS, max(1, (M * H + 63) // 64))
    part = torch.empty((4, nb, hd), dtype=F32, device=dqkv.device)
    _lib.check(_lib.load().vitm_qk_norm_bwd(_p(qkv_pre), qkv_pre.stride(0), _p(dqkv), dqkv.stride(0), _rm(qkv_pre, dqkv), M, H,
                                            hd, _p(_f32(q_w)), float(q_eps), _p(_f32(k_w)), float(k_eps), _p(part), nb,
                                            _stream()), "vitm_qk_norm_bwd")
    for i, out in enumerate((dqw_out, dqb_out, dkw_out, dkb_out)):
        colsum(part[i], nb, hd, out=out)


def gelu_bwd(pre, dh, dpre, M, D) -> None:
    _lib.check(_lib.load().vitm_gelu_bwd(_p(pre), pre.stride(0), _p(dh), dh.stride(0), _p(dpre), dpre.stride(0),
                                         _rm(pre, dh, dpre), M, D, _stream()), "vitm_gelu_bwd")


def layerscale_grads(weight, bias, dw, db, gamma, dgamma, round_bf16: bool) -> None:
    """dgamma of ``x + gamma * Linear(h)`` from the Linear's gradients taken without gamma (dw [N, K], db [N] or None, both
    contiguous fp32), which are then multiplied by gamma in place."""
    N, K = weight.shape
    assert dw.is_contiguous() and dw.shape == (N, K) and (bias is None) == (db is None)
    _lib.check(_lib.load().vitm_layerscale_grads(_p(_f32(weight.detach())), _p(None if bias is None else _f32(bias.detach())),
                                                 _p(_f32(dw)), _p(None if db is None else _f32(db)), _p(_f32(gamma)),
                                                 _p(_f32(dgamma)), N, K, 1 if round_bf16 else 0, _stream()),
               "vitm_layerscale_grads")


def residual_scale(resid, z, gamma, out, M, D) -> None:
    _lib.check(_lib.load().vitm_residual_scale(_p(resid), resid.stride(0), _p(z), z.stride(0), _p(_f32(gamma)), _p(out),
                                               out.stride(0), _rm(resid, z, out), M, D, _stream()), "vitm_residual_scale")


def cross_entropy(logits, labels, want_grad=True):
    """(loss scalar tensor, dlogits or None): mean softmax cross-entropy over the batch."""
    B, C = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    row_loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    dl = torch.empty((B, C), dtype=torch.float32, device=logits.device) if want_grad else None
    lab = labels.to(torch.int64).contiguous()
    _lib.check(_lib.load().vitm_cross_entropy_f32(_p(_f32(logits)), logits.stride(0), _p(lab), B, C, _p(loss), _p(dl),
                                                  C, _p(row_loss), _stream()), "vitm_cross_entropy_f32")
    return loss, dl


def grad_clip(flat_grads: torch.Tensor, max_norm: float) -> torch.Tensor:
    """In-place clip of one flat fp32 gradient buffer; returns a 2-element tensor (norm, coefficient)."""
    lib = _lib.load()
    scratch = torch.empty(lib.vitm_grad_clip_scratch_floats(), dtype=torch.float32, device=flat_grads.device)
    nc = torch.empty(2, dtype=torch.float32, device=flat_grads.device)
    _lib.check(lib.vitm_grad_clip_f32(_p(flat_grads), flat_grads.numel(), float(max_norm), _p(scratch), _p(nc), _stream()),
               "vitm_grad_clip_f32")
    return nc


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step) -> None:
    _lib.check(_lib.load().vitm_adamw_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                          float(eps), float(wd), int(step), _stream()), "vitm_adamw_f32")


def adamw_hyper_values(lr, beta1, beta2, eps, wd, step):
    """The seven floats vitm_adamw_f32_hyper reads, formed as vitm_adamw_f32 forms them from its arguments (betas rounded
    to fp32 first, bias corrections in double precision)."""
    import struct
    f32 = lambda x: struct.unpack("f", struct.pack("f", float(x)))[0]
    b1, b2 = f32(beta1), f32(beta2)
    return [float(lr), b1, b2, float(eps), float(wd), 1.0 - b1 ** int(step), (1.0 - b2 ** int(step)) ** 0.5]


def adamw_hyper(p, g, m, v, hyper) -> None:
    """AdamW update with (lr, beta1, beta2, eps, wd, 1 - beta1^t, sqrt(1 - beta2^t)) in the device tensor ``hyper``."""
    assert hyper.dtype == torch.float32 and hyper.numel() >= 7 and hyper.is_cuda
    _lib.check(_lib.load().vitm_adamw_f32_hyper(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _stream()),
               "vitm_adamw_f32_hyper")


# ---- Linear backward out of the forward GEMM ----------------------------------------------------------------------
def linear_dx(dy: torch.Tensor, weight: torch.Tensor, M: int, out: Optional[torch.Tensor] = None,
              gelu_input: Optional[torch.Tensor] = None, dy_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx[M, K] = dy[M, N] . W[N, K]  (nn.Linear weight layout): the forward GEMM with W^T as the packed operand.
    dy's dtype is the compute dtype (the fp32 master weight is packed to it); dx comes back in it.
    ``gelu_input`` (bf16 [M, >= K], the pre-activation of the GELU whose output fed this Linear) makes the epilogue
    multiply by GELU'(gelu_input): the result is the gradient wrt the pre-activation, with no separate pass.
    ``dy_scale`` (N floats) stands for dy * dy_scale (a LayerScale behind the Linear): it is folded into W^T."""
    N, K = weight.shape
    # W^T in the compute dtype, K of the GEMM (= N) padded to its multiple, rows (= K) padded to 8: one kernel
    kpad, npad = round_up(N, k_multiple()), round_up(K, 8)
    wt = transpose(weight.detach().contiguous(), N, K, kpad, out_dtype=dy.dtype, rows=npad, row_scale=dy_scale)
    pk = PackedLinear(wt, None, _zeros_f32(dy.device, npad), None, K, N, npad, kpad)
    dyp = dy
    if dy.shape[1] < pk.kpad:                                       # pad dy's columns to the GEMM's K multiple
        dyp = torch.zeros((M, pk.kpad), dtype=dy.dtype, device=dy.device)
        dyp[:, :dy.shape[1]] = dy
    if out is None:
        out = torch.empty((M, pk.npad), dtype=dy.dtype, device=dy.device)
    if gelu_input is not None:
        assert dy.dtype == torch.bfloat16 and gelu_input.dtype == torch.bfloat16 and gelu_input.shape[1] >= pk.npad
        gemm(dyp, pk, out, EPI_DGELU, M, resid=gelu_input)
    else:
        gemm(dyp, pk, out, EPI_BIAS, M)
    return out


def _tn_ok(t: torch.Tensor) -> bool:
    return t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0


def gemm_tn(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, out: Optional[torch.Tensor] = None,
            colsum_out: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[N, K] (fp32) = a[:M, :N]^T . b[:M, :K] for bf16 row-major a, b: no transposed copies, reduction split over
    workgroup slices and summed in a fixed order.  ``colsum_out`` (N floats) also receives the column sums of a;
    ``row_scale`` (N floats) multiplies row n of both results."""
    lib = _lib.load()
    if out is None:
        out = torch.empty((N, K), dtype=F32, device=a.device)
    assert out.stride(1) == 1 and (colsum_out is None or (colsum_out.is_contiguous() and colsum_out.numel() == N))
    nws = lib.vitm_gemm_tn_workspace_floats(M, N, K)
    ws = torch.empty(nws, dtype=F32, device=a.device) if nws else None
    _lib.check(lib.vitm_gemm_tn(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), _p(colsum_out),
                                _p(None if row_scale is None else _f32(row_scale)), M, N, K, _p(ws), nws, _stream()),
               "vitm_gemm_tn")
    return out


def linear_dw(dy: torch.Tensor, x: torch.Tensor, M: int, N: int, K: int, out: Optional[torch.Tensor] = None,
              bias_out: Optional[torch.Tensor] = None, dy_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[N, K] (fp32) = dy[:, :N]^T . x[:, :K], and the bias gradient sum_m dy[m, :N] into ``bias_out`` when given.
    bf16 operands: the transposing weight-gradient GEMM (fp32 accumulators stored as fp32, the bias gradient from the
    same pass).  fp32 operands: the forward GEMM on the two transposed activations (rows padded to its K multiple) and
    a column-sum kernel.  ``out`` (contiguous [N, K] fp32) receives dW when given.  ``dy_scale`` (N floats, bf16 path
    only) stands for dy * dy_scale."""
    if dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and _tn_ok(dy) and _tn_ok(x):
        return gemm_tn(dy, x, M, N, K, out, bias_out, dy_scale)
    assert dy_scale is None, "dy_scale needs the bf16 weight-gradient GEMM"
    if bias_out is not None:
        colsum(dy, M, N, out=bias_out)
    mp = round_up(M, k_multiple())
    dyt = transpose(dy, M, N, mp)                                   # [N, mp]
    xt = transpose(x, M, K, mp)                                     # [K, mp]
    pk = pack_linear(xt, None, dtype=dy.dtype)                      # "weight" = x^T: rows K, cols mp
    if out is not None and pk.npad == K:
        gemm(dyt, pk, out, EPI_BIAS, N)
        return out
    res = torch.empty((N, pk.npad), dtype=F32, device=dy.device)
    gemm(dyt, pk, res, EPI_BIAS, N)
    if out is not None:
        out.copy_(res[:, :K])
        return out
    return res[:, :K]
