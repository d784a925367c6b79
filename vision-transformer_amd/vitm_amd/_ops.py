"""Torch-tensor front end of the C ABI: argument checking, packing caches and launch helpers.

PyTorch supplies device memory and the current stream; every computation below is a call into
libvitm_gfx950.so.  Nothing here computes with torch ops on the data path.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import BF16, F32, EPI_BIAS, EPI_BIAS_GELU, EPI_LN, EPI_LN_GELU, EPI_RESIDUAL, EPI_DGELU, EPI_BIAS_GELU2  # noqa: F401


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def k_multiple() -> int:
    return _lib.load().vitm_gemm_k_multiple()


def _p(t: Optional[torch.Tensor]):
    """Device address for a c_void_p parameter (ctypes converts a plain int; no wrapper object per argument)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current device's current stream as a hipStream_t.  torch.cuda.current_stream() costs several microseconds
    of Python per call -- a third of the host time of a launch-bound forward -- so the C accessors are used when this
    torch has them."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(
            f"{what}: vitm_amd computes only on an MI355X through libvitm_gfx950 -- move the module and its "
            f"input to a 'cuda' device (got a {x.device} tensor; there is no CPU path)")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype} (bf16 or fp32)")


def f32vec(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A parameter vector as a contiguous fp32 tensor: no copy when it already is one, otherwise a copy that is
    memoised on the tensor until its storage or version changes (so a bf16 module converts once, not per call)."""
    if t is None:
        return None
    d = t.detach()
    if d.dtype == torch.float32 and d.is_contiguous():
        return d
    key = (t.data_ptr(), t._version, t.dtype, t.device, tuple(t.shape))
    hit = getattr(t, "_vitm_f32", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = d.to(torch.float32).contiguous()
    try:
        t._vitm_f32 = (key, val)
    except AttributeError:  # pragma: no cover - tensors that refuse attributes just convert every time
        pass
    return val


class PackedLinear:
    """Packed weight [npad, kpad] (bf16 for vitm_gemm, fp32 for vitm_gemm_f32) plus the folded LayerNorm /
    LayerScale vectors."""
    __slots__ = ("wp", "s", "c", "g", "N", "K", "npad", "kpad")

    def __init__(self, wp, s, c, g, N, K, npad, kpad):
        self.wp, self.s, self.c, self.g = wp, s, c, g
        self.N, self.K, self.npad, self.kpad = N, K, npad, kpad

    @property
    def dtype(self):
        return self.wp.dtype


def compute_dtype(x: torch.Tensor) -> torch.dtype:
    """The arithmetic path a tensor takes: fp32 tensors run the fp32 kernels, bf16 tensors the bf16 ones."""
    if x.dtype in (torch.float32, torch.bfloat16):
        return x.dtype
    raise TypeError(f"unsupported activation dtype {x.dtype} (bf16 or fp32)")


def pack_linear(weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                ln_weight: Optional[torch.Tensor] = None, ln_bias: Optional[torch.Tensor] = None,
                out_scale: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.bfloat16) -> PackedLinear:
    """Run vitm_prepack_linear on an (N, K...) weight; N is padded to 8 and K to the GEMM's K multiple."""
    lib = _lib.load()
    w = weight.detach()
    if w.dtype not in (torch.bfloat16, torch.float32):
        w = w.float()
    w = w.reshape(w.shape[0], -1).contiguous()
    N, K = w.shape
    npad, kpad = round_up(N, 8), round_up(K, k_multiple())
    dev = w.device
    # the kernel writes rows 0 .. N-1 in full (K padding included); only the N padding needs clearing
    fold_ln = ln_weight is not None
    alloc = torch.empty if npad == N else torch.zeros
    wp = alloc((npad, kpad), dtype=dtype, device=dev)
    s = alloc(npad, dtype=torch.float32, device=dev) if fold_ln else None
    c = alloc(npad, dtype=torch.float32, device=dev)
    g = alloc(npad, dtype=torch.float32, device=dev) if out_scale is not None else None
    b32, lw32, lb32, os32 = f32vec(bias), f32vec(ln_weight), f32vec(ln_bias), f32vec(out_scale)
    if fold_ln and lb32 is None:
        lb32 = torch.zeros_like(lw32)
    _lib.check(lib.vitm_prepack_linear(_p(w), _dt(w), _p(b32), _p(lw32), _p(lb32), _p(os32), N, K, kpad,
                                       _p(wp), _dt(wp), _p(s), _p(c), _p(g), _stream()), "vitm_prepack_linear")
    return PackedLinear(wp, s, c, g, N, K, npad, kpad)


def cached_pack(owner, slot: str, tensors: Sequence[Optional[torch.Tensor]], build):
    """Memoise a packed weight on ``owner`` until one of the source tensors changes."""
    key = tuple((t.data_ptr(), t._version, t.dtype, t.device) if t is not None else None for t in tensors)
    cache = owner.__dict__.setdefault("_vitm_packs", {})
    hit = cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = build()
    cache[slot] = (key, val)
    return val


def ln_stats(x: torch.Tensor, M: int, D: int, eps: float, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x is a 2-D (rows, ld) view; returns [M, 2] fp32 (mean, rstd)."""
    if stats is None:
        stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vitm_ln_stats(_p(x), _dt(x), x.stride(0), M, D, float(eps), _p(stats), _stream()),
               "vitm_ln_stats")
    return stats


def uses_big_tile(M: int, pk: "PackedLinear") -> bool:
    """True when vitm_gemm runs (M, pk) on the persistent 256x256-tile kernel (bf16 only)."""
    return pk.dtype == torch.bfloat16 and bool(_lib.load().vitm_gemm_uses_big_tile(M, pk.npad, pk.kpad))


def new_carried_stats(M: int, device) -> torch.Tensor:
    """[M, VITM_STATS_PARTS, 2] buffer for the LayerNorm moments that vitm_gemm_stats carries between GEMMs.  It is
    part of a forward's own workspace (never shared between forwards, so graph replays and forwards on several
    streams cannot meet in it) and needs no clearing: the producing epilogue writes the slots of its column tiles
    and zero-fills the slots of the column tiles the shape does not have, and a consumer only ever reads what a
    producer of the same forward wrote."""
    return torch.empty((M, _lib.STATS_PARTS, 2), dtype=torch.float32, device=device)


def gemm(A: torch.Tensor, pk: PackedLinear, out: torch.Tensor, epilogue: int, M: int,
         stats: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, row_group: int = 0,
         stats_out: Optional[torch.Tensor] = None, stats_partial: bool = False, ln_eps: float = 0.0) -> torch.Tensor:
    """out[:, :npad] = epilogue(A[:, :kpad] . wp^T); A, out (and resid) are 2-D row-major views.
    fp32 A runs the fp32 kernel (fp32 packed weight, fp32 out and resid).
    ``stats_out`` (RESIDUAL epilogue) / ``stats_partial`` (LN epilogues, with ``ln_eps``) carry the LayerNorm row
    moments between GEMMs (vitm_gemm_stats; big-tile shapes only, see ``uses_big_tile``)."""
    assert A.stride(1) == 1 and out.stride(1) == 1
    assert A.shape[1] >= pk.kpad, "A operand narrower than the packed K"
    assert A.dtype == pk.dtype, f"A is {A.dtype} but the weight was packed as {pk.dtype}"
    if A.dtype == torch.float32:
        assert out.dtype == torch.float32 and (resid is None or resid.dtype == torch.float32)
        _lib.check(_lib.load().vitm_gemm_f32(
            _p(A), A.stride(0), _p(pk.wp), pk.kpad, _p(out), out.stride(0), M, pk.npad, pk.kpad, epilogue,
            _p(pk.c), _p(pk.g), _p(resid), 0 if resid is None else resid.stride(0), row_group, _stream()),
            "vitm_gemm_f32")
        return out
    assert A.dtype == torch.bfloat16
    if stats_out is not None or stats_partial:
        assert row_group == 0 and out.dtype == torch.bfloat16
        _lib.check(_lib.load().vitm_gemm_stats(
            _p(A), A.stride(0), _p(pk.wp), pk.kpad, _p(out), out.stride(0), M, pk.npad, pk.kpad, epilogue,
            _p(stats), _p(pk.s), _p(pk.c), _p(pk.g), _p(resid), 0 if resid is None else resid.stride(0),
            _p(stats_out), 1 if stats_partial else 0, pk.K, float(ln_eps), _stream()), "vitm_gemm_stats")
        return out
    _lib.check(_lib.load().vitm_gemm(
        _p(A), A.stride(0), _p(pk.wp), pk.kpad, _p(out), out.stride(0), _dt(out), M, pk.npad, pk.kpad, epilogue,
        _p(stats), _p(pk.s), _p(pk.c), _p(pk.g), _p(resid), 0 if resid is None else resid.stride(0), row_group,
        _stream()), "vitm_gemm")
    return out


def im2col(img: torch.Tensor, A: torch.Tensor, patch, kpad: int, cls_slot: bool) -> None:
    """Patch matrix of an NCHW batch into A [rows, kpad] (bf16 or fp32); class-token rows (cls_slot) and K padding zero."""
    B, C, H, W = img.shape
    _lib.check(_lib.load().vitm_im2col(_p(img), _dt(img), B, C, H, W, patch[0], patch[1], _p(A), _dt(A), A.stride(0), kpad,
                                       1 if cls_slot else 0, _stream()), "vitm_im2col")


def patch_embed(img: torch.Tensor, pk: PackedLinear, tokens: torch.Tensor, patch, D: int, cls_slot: bool) -> None:
    B, C, H, W = img.shape
    # bf16 images with a power-of-two patch width need no patch matrix: at large batch vitm_patch_embed gathers the
    # patches inside the big-tile kernel's loader
    pw = patch[1]
    gathered = (img.dtype == torch.bfloat16 and pw & (pw - 1) == 0 and 8 <= pw <= 64 and W % 8 == 0
                and pk.kpad == C * patch[0] * pw and img.data_ptr() % 32 == 0)
    if tokens.dtype == torch.bfloat16 and pk.npad == D and not gathered:
        # large batch otherwise: one pass that writes the bf16 patch matrix, then the persistent big-tile GEMM
        npatch = (H // patch[0]) * (W // patch[1])
        rows = B * (npatch + (1 if cls_slot else 0))
        lib = _lib.load()
        if lib.vitm_gemm_uses_big_tile(rows, pk.npad, pk.kpad):
            A = torch.empty((rows, pk.kpad), dtype=torch.bfloat16, device=img.device)
            im2col(img, A, patch, pk.kpad, cls_slot)
            gemm(A, pk, tokens, EPI_BIAS, rows)
            return
    if tokens.dtype == torch.float32:
        assert pk.dtype == torch.float32
        _lib.check(_lib.load().vitm_patch_embed_f32(
            _p(img), _dt(img), B, C, H, W, patch[0], patch[1], _p(pk.wp), pk.kpad, pk.kpad, _p(pk.c), _p(tokens),
            tokens.stride(0), D, 1 if cls_slot else 0, _stream()), "vitm_patch_embed_f32")
        return
    _lib.check(_lib.load().vitm_patch_embed(
        _p(img), _dt(img), B, C, H, W, patch[0], patch[1], _p(pk.wp), pk.kpad, pk.kpad, _p(pk.c), _p(tokens),
        tokens.stride(0), D, 1 if cls_slot else 0, _stream()), "vitm_patch_embed")


def embed_finalize(x: torch.Tensor, B: int, ntok: int, D: int, cls, pos, en_w, en_b, en_eps, np_w, np_b, np_eps,
                   stats: Optional[torch.Tensor] = None, stats_eps: float = 0.0) -> None:
    """``stats`` ([B*ntok, 2] fp32): also leave (mean, rstd) of the finished rows, as vitm_ln_stats would."""
    _lib.check(_lib.load().vitm_embed_finalize_stats(
        _p(x), _dt(x), x.stride(0), B, ntok, D, _p(cls), _p(pos), _p(en_w), _p(en_b), float(en_eps), _p(np_w), _p(np_b),
        float(np_eps), _p(stats), float(stats_eps), _stream()), "vitm_embed_finalize")


def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, N: int, H: int, hd: int, scale: float) -> None:
    assert qkv.dtype == out.dtype
    if qkv.dtype == torch.float32:
        _lib.check(_lib.load().vitm_attention_f32(_p(qkv), qkv.stride(0), _p(out), out.stride(0), B, N, H, hd,
                                                  float(scale), _stream()), "vitm_attention_f32")
        return
    _lib.check(_lib.load().vitm_attention(_p(qkv), qkv.stride(0), _p(out), out.stride(0), B, N, H, hd, float(scale),
                                          _stream()), "vitm_attention")


def qk_norm(qkv: torch.Tensor, M: int, H: int, hd: int, qw, qb, qeps: float, kw, kb, keps: float) -> None:
    _lib.check(_lib.load().vitm_qk_norm(_p(qkv), _dt(qkv), qkv.stride(0), M, H, hd, _p(qw), _p(qb), float(qeps),
                                        _p(kw), _p(kb), float(keps), _stream()), "vitm_qk_norm")


def layernorm(x: torch.Tensor, out: torch.Tensor, M: int, D: int, w: torch.Tensor, b: torch.Tensor, eps: float) -> None:
    _lib.check(_lib.load().vitm_layernorm(_p(x), _dt(x), x.stride(0), _p(out), _dt(out), out.stride(0), M, D, _p(w), _p(b),
                                          float(eps), _stream()), "vitm_layernorm")


def scale_channels(x: torch.Tensor, out: torch.Tensor, M: int, D: int, gamma: torch.Tensor) -> None:
    assert x.dtype == out.dtype
    _lib.check(_lib.load().vitm_scale_channels(_p(x), x.stride(0), _p(out), out.stride(0), _dt(x), M, D, _p(gamma),
                                               _stream()), "vitm_scale_channels")


def gelu(x: torch.Tensor, out: torch.Tensor, M: int, D: int) -> None:
    assert x.dtype == out.dtype
    _lib.check(_lib.load().vitm_gelu(_p(x), x.stride(0), _p(out), out.stride(0), _dt(x), M, D, _stream()), "vitm_gelu")


def activation_kind(act) -> Optional[int]:
    """enum vitm_activation_kind of an activation module, or None when the library has no kernel for it."""
    import torch.nn as nn
    if isinstance(act, nn.GELU):
        approx = getattr(act, "approximate", "none")
        return _lib.ACT_GELU_ERF if approx == "none" else (_lib.ACT_GELU_TANH if approx == "tanh" else None)
    for cls, kind in ((nn.ReLU, _lib.ACT_RELU), (nn.SiLU, _lib.ACT_SILU), (nn.Sigmoid, _lib.ACT_SIGMOID), (nn.Tanh, _lib.ACT_TANH)):
        if type(act) is cls:
            return kind
    return None


def activation(x: torch.Tensor, out: torch.Tensor, M: int, D: int, kind: int) -> None:
    assert x.dtype == out.dtype
    _lib.check(_lib.load().vitm_activation(_p(x), x.stride(0), _p(out), out.stride(0), _dt(x), M, D, kind, _stream()),
               "vitm_activation")


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """A contiguous bf16 copy of a contiguous fp32/bf16 CUDA tensor (cast kernel, not torch)."""
    x = x.contiguous()
    if x.dtype == torch.bfloat16:
        return x
    if x.dtype != torch.float32:
        raise TypeError(f"unsupported activation dtype {x.dtype} (bf16 or fp32)")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    if x.numel():
        _lib.check(_lib.load().vitm_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "vitm_cast_f32_to_bf16")
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    if x.dtype == torch.float32:
        return x
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if x.numel():
        _lib.check(_lib.load().vitm_cast_bf16_to_f32(_p(x), _p(out), x.numel(), _stream()), "vitm_cast_bf16_to_f32")
    return out


def like_input(y: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """Return a result in the caller's dtype (a cast kernel only when the two differ)."""
    if y.dtype == ref.dtype:
        return y
    return to_f32(y) if ref.dtype == torch.float32 else to_bf16(y)
