"""Per-kernel parity on a real MI355X: each C-ABI entry point against a plain fp32 PyTorch statement of the
same op on the same (bf16-rounded) inputs.

Tolerances.  Outputs are bf16 (relative spacing 2^-8 = 3.9e-3) of an fp32-accumulated result, so the bar is
|got - ref| <= 1e-2 * max|ref| for GEMM-shaped results (half an ulp of the largest output, plus slack for the
reference's own rounding) and 2e-2 absolute for O(1) normalised outputs.  fp32 outputs: 1e-4 relative.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from vitm_amd import _ops
    return _ops


def _bf(x):
    return x.to(torch.bfloat16)


def _close(got, ref, rel=1e-2, what=""):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 256, 256), (200, 136, 128), (1, 8, 64),
                                   (197 * 3, 768, 768), (1000, 2304, 768), (333, 616, 192)])
def test_gemm_bias(ops, M, N, K):
    A = _bf(_rand(M, K, seed=1))
    W = _rand(N, K, seed=2, scale=K ** -0.5)
    b = _rand(N, seed=3)
    pk = ops.pack_linear(W, b)
    assert pk.kpad == K and pk.npad == N
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, pk, out, ops.EPI_BIAS, M)
    ref = A.float() @ _bf(W).float().t() + b
    _close(out, ref, what="gemm bias")


@pytest.mark.parametrize("epi", ["bias", "ln", "ln_gelu", "resid", "bias_f32"])
@pytest.mark.parametrize("M,N,K", [(8200, 2304, 768), (11111, 768, 3072), (9000, 1152, 384), (33000, 3072, 768),
                                   (16400, 520, 64), (50432, 768, 768), (50432, 3072, 768), (40000, 1024, 128),
                                   (66000, 512, 256), (50432, 384, 384), (33333, 384, 1536), (40001, 320, 128),
                                   (70000, 264, 64)])
def test_gemm_big_tile_kernel(ops, M, N, K, epi):
    """Shapes large enough to be routed to the 256x256 ping-pong kernel (ragged M and N tails included); shapes 5-9
    leave a partly filled last round whose tiles are shared 3, 4, 2 and 4 ways between workgroups; the last four have a
    second column tile at most 128 columns wide (N = 384: ViT-S/16), which runs as narrow tiles (128, 64 and 8 columns)."""
    A = _bf(_rand(M, K, seed=40) * 1.3 + 0.2)
    W, b = _rand(N, K, seed=41, scale=K ** -0.5), _rand(N, seed=42, scale=0.2)
    F = torch.nn.functional
    if epi in ("ln", "ln_gelu"):
        lw, lb = 1.0 + 0.1 * _rand(K, seed=43), 0.1 * _rand(K, seed=44)
        pk = ops.pack_linear(W, b, lw, lb)
        stats = ops.ln_stats(A, M, K, 1e-5)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.gemm(A, pk, out, ops.EPI_LN_GELU if epi == "ln_gelu" else ops.EPI_LN, M, stats=stats)
        ref = F.layer_norm(A.float(), (K,), lw, lb, 1e-5) @ W.t() + b
        if epi == "ln_gelu":
            ref = F.gelu(ref)
        _close(out, ref, rel=1.5e-2, what=f"big {epi}")
    elif epi == "resid":
        gamma = 1.0 + 0.1 * _rand(N, seed=45)
        x = _bf(_rand(M, N, seed=46))
        ref = x.float() + gamma * (A.float() @ _bf(W).float().t() + b)
        pk = ops.pack_linear(W, b, out_scale=gamma)
        ops.gemm(A, pk, x, ops.EPI_RESIDUAL, M, resid=x)
        _close(x, ref, what="big resid")
    else:
        pk = ops.pack_linear(W, b)
        dt = torch.float32 if epi == "bias_f32" else torch.bfloat16
        out = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
        ops.gemm(A, pk, out, ops.EPI_BIAS, M)
        ref = A.float() @ _bf(W).float().t() + b
        _close(out, ref, rel=1e-4 if epi == "bias_f32" else 1e-2, what=f"big {epi}")


@pytest.mark.parametrize("M,N,K", [(300, 200, 128), (788, 3072, 768), (8200, 2304, 768), (50432, 3072, 768), (33333, 1536, 384),
                                   (16400, 520, 64), (70000, 264, 64)])
def test_gemm_two_outputs_pre_activation_and_gelu(ops, M, N, K):
    """VITM_EPI_BIAS_GELU2 (the training forward's fc1): `out` = the EPI_BIAS result bit for bit, the second output its
    GELU; on the 128x128 route (first two shapes) and the 256x256 route incl. ragged edges, shared leftover tiles and a
    narrow last column tile.  Pad columns / rows outside (M, N) of both buffers stay untouched."""
    A = _bf(_rand(M, K, seed=60) * 1.3 + 0.2)
    W, b = _rand(N, K, seed=61, scale=K ** -0.5), _rand(N, seed=62, scale=0.2)
    pk = ops.pack_linear(W, b)
    ld = pk.npad + 8
    want = torch.empty((M, pk.npad), dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, pk, want, ops.EPI_BIAS, M)
    pre = torch.full((M + 1, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    act = torch.full((M + 1, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, pk, pre, ops.EPI_BIAS_GELU2, M, resid=act)
    assert torch.equal(pre[:M, :pk.npad], want), "pre-activation differs from the plain bias epilogue"
    ref = torch.nn.functional.gelu(A.float() @ _bf(W).float().t() + b)
    _close(act[:M, :N], ref, rel=1e-2, what="gelu output")
    # one bf16 ulp around gelu(bf16 pre-activation): the GELU is taken from the fp32 value before rounding
    near = torch.nn.functional.gelu(want.float())
    assert ((act[:M, :pk.npad].float() - near).abs() <= 2.0 ** -7 * near.abs() + 1e-3).all()
    for buf in (pre, act):
        assert (buf[M] == 7.0).all() and (buf[:, pk.npad:] == 7.0).all(), "wrote outside the result"


@pytest.mark.parametrize("M,N,K,Kn", [(50432, 768, 768, 2304), (50432, 768, 3072, 3072), (33000, 1024, 256, 1024),
                                      (40000, 384, 128, 1152), (66000, 512, 256, 512), (16400, 520, 64, 1040),
                                      (50432, 384, 1536, 1536), (33333, 384, 384, 1152)])
def test_gemm_carried_layernorm_stats(ops, M, N, K, Kn):
    """The RESIDUAL epilogue's row moments (vitm_gemm_stats stats_out) against float64 moments of the stream it wrote,
    and an LN-folded GEMM fed with them (stats_partial) against the same GEMM fed by vitm_ln_stats.  Shapes cover
    1-4 column tiles, a ragged last column tile (N = 384, 520), ragged M and leftover tiles shared 2, 3 and 4 ways."""
    A = _bf(_rand(M, K, seed=50))
    W, b = _rand(N, K, seed=51, scale=K ** -0.5), _rand(N, seed=52, scale=0.2)
    gamma = 1.0 + 0.1 * _rand(N, seed=53)
    x = _bf(_rand(M, N, seed=54) * 1.5 + 0.7)          # mean well away from 0: the raw moments must cope
    pk = ops.pack_linear(W, b, out_scale=gamma)
    assert ops.uses_big_tile(M, pk)
    carried = ops.new_carried_stats(M, DEV)
    carried[:] = float("nan")      # every slot must be overwritten: the shape's tiles with moments, the others with zeros
    ops.gemm(A, pk, x, ops.EPI_RESIDUAL, M, resid=x, stats_out=carried)
    xs = x.double()
    tot = carried.double().sum(1)
    assert torch.isfinite(tot).all()
    mean, var = tot[:, 0] / N, tot[:, 1] / N - (tot[:, 0] / N) ** 2
    rmean, rvar = xs.mean(1), xs.var(1, unbiased=False)
    # the moments are those of the fp32 values before their rounding to bf16: the mean moves by << 1e-3 sigma
    assert ((mean - rmean).abs() / rvar.sqrt()).max().item() < 1e-3
    assert ((var - rvar).abs() / rvar).max().item() < 2e-3
    # consumer: LayerNorm over the N columns of x folded into the next GEMM
    lw, lb = 1.0 + 0.1 * _rand(N, seed=55), 0.1 * _rand(N, seed=56)
    W2, b2 = _rand(Kn, N, seed=57, scale=N ** -0.5), _rand(Kn, seed=58, scale=0.2)
    pk2 = ops.pack_linear(W2, b2, lw, lb)
    if pk2.kpad != N:
        return   # N = 520 has K padding in the consumer; the moments above are what this case is for
    assert ops.uses_big_tile(M, pk2)
    out_a = torch.full((M, Kn), float("nan"), dtype=torch.bfloat16, device=DEV)
    out_b = torch.full((M, Kn), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, pk2, out_a, ops.EPI_LN, M, stats=carried, stats_partial=True, ln_eps=1e-5)
    ops.gemm(x, pk2, out_b, ops.EPI_LN, M, stats=ops.ln_stats(x, M, N, 1e-5))
    ref = torch.nn.functional.layer_norm(x.float(), (N,), lw, lb, 1e-5) @ W2.t() + b2
    _close(out_a, ref, rel=1.5e-2, what="LN GEMM on carried moments")
    _close(out_a, out_b, rel=8e-3, what="carried moments vs vitm_ln_stats")


def test_gemm_identity_asymmetric(ops):
    """A = I against an asymmetric W catches a transposed / permuted fragment map exactly."""
    K = N = 128
    A = _bf(torch.eye(K, device=DEV))
    W = (torch.arange(N, device=DEV).float()[:, None] * 0.5 + torch.arange(K, device=DEV).float()[None, :] * 0.0078125)
    pk = ops.pack_linear(W, None)
    out = torch.empty((K, N), dtype=torch.float32, device=DEV)
    ops.gemm(A, pk, out, ops.EPI_BIAS, K)
    ref = _bf(W).float().t()
    assert torch.equal(out, ref)


def test_gemm_k_padding_and_f32_out(ops):
    M, N, K = 77, 10, 100  # K -> 128, N -> 16
    A = torch.zeros((M, 128), dtype=torch.bfloat16, device=DEV)
    A[:, :K] = _bf(_rand(M, K, seed=4))
    W, b = _rand(N, K, seed=5, scale=0.1), _rand(N, seed=6)
    pk = ops.pack_linear(W, b)
    assert (pk.kpad, pk.npad) == (128, 16)
    out = torch.empty((M, 16), dtype=torch.float32, device=DEV)
    ops.gemm(A, pk, out, ops.EPI_BIAS, M)
    ref = A[:, :K].float() @ _bf(W).float().t() + b
    _close(out[:, :N], ref, rel=1e-4, what="gemm f32 out")
    assert (out[:, N:] == 0).all()


@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("M,N,D", [(197 * 2, 2304, 768), (130, 512, 128), (65, 192 * 3, 192)])
def test_gemm_layernorm_folded(ops, M, N, D, gelu):
    x = _bf(_rand(M, D, seed=7) * 1.5 + 0.3)
    W, b = _rand(N, D, seed=8, scale=D ** -0.5), _rand(N, seed=9, scale=0.1)
    lw, lb = 1.0 + 0.1 * _rand(D, seed=10), 0.1 * _rand(D, seed=11)
    pk = ops.pack_linear(W, b, lw, lb)
    stats = ops.ln_stats(x, M, D, 1e-5)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    ops.gemm(x, pk, out, ops.EPI_LN_GELU if gelu else ops.EPI_LN, M, stats=stats)
    xn = torch.nn.functional.layer_norm(x.float(), (D,), lw, lb, 1e-5)
    ref = xn @ W.t() + b
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    # the folded form rounds W*gamma (not LN(x)) to bf16: same error class as a bf16 linear
    _close(out, ref, rel=1.5e-2, what="ln-folded gemm")


def test_ln_stats(ops):
    M, D = 300, 768
    x = _bf(_rand(M, D, seed=12) * 2 + 5.0)
    st = ops.ln_stats(x, M, D, 1e-5)
    xf = x.float()
    mean = xf.mean(-1)
    rstd = 1.0 / torch.sqrt(xf.var(-1, unbiased=False) + 1e-5)
    _close(st[:, 0], mean, rel=1e-5, what="mean")
    _close(st[:, 1], rstd, rel=1e-4, what="rstd")


@pytest.mark.parametrize("M,N,K", [(197 * 2, 768, 3072), (100, 128, 512), (50, 192, 768)])
def test_gemm_residual_layerscale_inplace(ops, M, N, K):
    A = _bf(_rand(M, K, seed=13))
    W, b = _rand(N, K, seed=14, scale=K ** -0.5), _rand(N, seed=15, scale=0.1)
    gamma = 1.0 + 0.1 * _rand(N, seed=16)
    x = _bf(_rand(M, N, seed=17))
    ref = x.float() + gamma * (A.float() @ _bf(W).float().t() + b)
    pk = ops.pack_linear(W, b, out_scale=gamma)
    ops.gemm(A, pk, x, ops.EPI_RESIDUAL, M, resid=x)  # in place on the residual stream
    _close(x, ref, what="residual gemm")


def test_gelu_epilogue_matches_erf_gelu(ops):
    M, N, K = 128, 128, 64
    A = _bf(_rand(M, K, seed=18) * 2)
    W = _rand(N, K, seed=19, scale=0.3)
    pk = ops.pack_linear(W, None)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    ops.gemm(A, pk, out, ops.EPI_BIAS_GELU, M)
    ref = torch.nn.functional.gelu(A.float() @ _bf(W).float().t())
    _close(out, ref, what="gelu")


def _attn_ref(qkv, B, N, H, hd):
    D = H * hd
    q, k, v = qkv.float().reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q * hd ** -0.5) @ k.transpose(-2, -1)
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)


@pytest.mark.parametrize("B,N,H,hd", [(2, 197, 3, 64), (1, 65, 2, 64), (1, 577, 2, 64), (3, 32, 1, 64),
                                      (2, 1, 2, 64), (1, 65, 2, 96), (1, 197, 2, 32), (1, 100, 1, 128),
                                      (1, 300, 1, 64),
                                      # both sides of the kernel boundaries at head dim 64 (224 | 225: one pair's keys in
                                      # the ring | resident keys; 608 | 609: resident | chunked keys), longer sequences,
                                      # and more (image, head) pairs than workgroups (several items per workgroup)
                                      (1, 224, 2, 64), (1, 225, 2, 64), (1, 608, 1, 64), (1, 609, 1, 64), (1, 785, 2, 64),
                                      (1, 1025, 1, 64), (30, 197, 12, 64), (25, 577, 12, 64), (12, 640, 12, 64),
                                      # head dims the kernels pad to the next multiple of 32 (80: ViT-H/14)
                                      (2, 257, 3, 80), (1, 65, 2, 48), (1, 197, 2, 8), (1, 50, 3, 40), (1, 130, 2, 104),
                                      (1, 70, 1, 120)])
def test_attention(ops, B, N, H, hd):
    D = H * hd
    qkv = _bf(_rand(B * N, 3 * D, seed=20) * 1.5)
    out = torch.full((B * N, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.attention(qkv, out, B, N, H, hd, hd ** -0.5)
    _close(out, _attn_ref(qkv, B, N, H, hd), rel=1.5e-2, what="attention")


def test_attention_online_softmax_rescale(ops):
    """A late key far above all earlier scores forces the running-max rescale branch in every tile order."""
    B, N, H, hd = 1, 197, 1, 64
    qkv = _bf(_rand(B * N, 3 * hd, seed=21))
    q = qkv[:, :hd].float()
    for key_row in (5, 100, 196):  # first tile, middle tile, ragged last tile
        t = qkv.clone()
        t[key_row, hd:2 * hd] = _bf(q[7] * 6.0)  # k[key_row] aligned with q[7]: its score dominates row 7
        out = torch.empty((N, hd), dtype=torch.bfloat16, device=DEV)
        ops.attention(t, out, B, N, H, hd, hd ** -0.5)
        _close(out, _attn_ref(t, B, N, H, hd), rel=1.5e-2, what=f"attention spike at key {key_row}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("img,p,D,B", [(224, 16, 768, 2), (32, 4, 192, 3), (32, 8, 128, 2), (28, 14, 64, 1)])
def test_patch_embed(ops, img, p, D, B, dtype):
    from oracle import vit_oracle as O
    x = (_rand(B, 3, img, img, seed=22)).to(dtype)
    w = _rand(D, 3, p, p, seed=23, scale=(3 * p * p) ** -0.5)
    b = _rand(D, seed=24, scale=0.1)
    pk = ops.pack_linear(w, b)
    g = img // p
    Np = g * g
    for cls_slot in (False, True):
        rows = B * (Np + int(cls_slot))
        tok = torch.full((rows, D), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.patch_embed(x, pk, tok, (p, p), D, cls_slot)
        cols = O.im2col(x.float().cpu(), (p, p)).to(DEV)
        ref = cols @ _bf(w).float().reshape(D, -1).t() + b
        got = tok.view(B, Np + int(cls_slot), D)
        if cls_slot:
            assert (got[:, 0] == 7.0).all(), "class-token slot must be left untouched"
            got = got[:, 1:]
        # fp32 images are rounded to bf16 on the way into the MFMA
        ref_in = O.im2col(x.to(torch.bfloat16).float().cpu(), (p, p)).to(DEV) @ _bf(w).float().reshape(D, -1).t() + b
        _close(got, ref_in, what="patch embed")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("img,p,D,B", [(224, 16, 768, 160), (64, 8, 256, 520), (60, 6, 256, 400)])
def test_patch_embed_large_batch_route(ops, img, p, D, B, dtype):
    """Batches large enough for the big-tile GEMM: bf16 images with a power-of-two patch width are gathered by the
    kernel's own loader, everything else goes through the patch-matrix pass (vector and element-wise gathers)."""
    from oracle import vit_oracle as O
    from vitm_amd import _lib
    x = (_rand(B, 3, img, img, seed=60)).to(dtype)
    w = _rand(D, 3, p, p, seed=61, scale=(3 * p * p) ** -0.5)
    b = _rand(D, seed=62, scale=0.1)
    pk = ops.pack_linear(w, b)
    Np = (img // p) ** 2
    assert _lib.load().vitm_gemm_uses_big_tile(B * Np, pk.npad, pk.kpad) == 1
    for cls_slot in (False, True):
        rows = B * (Np + int(cls_slot))
        tok = torch.full((rows, D), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.patch_embed(x, pk, tok, (p, p), D, cls_slot)
        got = tok.view(B, Np + int(cls_slot), D)
        if cls_slot:
            if dtype == torch.bfloat16 and p & (p - 1) == 0:   # gathered inside the big-tile kernel: the slot is not written
                assert (got[:, 0] == 7.0).all(), "class-token slot must be left untouched"
            else:                         # patch-matrix route: the slot's patch row is zero, so it holds the bias
                _close(got[:, 0], b.expand(B, D), what="class-token slot holds the bias until embed_finalize")
            got = got[:, 1:]
        for lo in range(0, B, 64):   # chunked reference
            cols = O.im2col(x[lo:lo + 64].to(torch.bfloat16).float().cpu(), (p, p)).to(DEV)
            ref = cols @ _bf(w).float().reshape(D, -1).t() + b
            _close(got[lo:lo + 64], ref, what="patch embed (large batch)")
        if dtype == torch.bfloat16 and p & (p - 1) == 0:
            # the gathering loader feeds the same kernel the same operands in the same order as the patch-matrix route
            A = torch.empty((rows, pk.kpad), dtype=torch.bfloat16, device=DEV)
            ops.im2col(x, A, (p, p), pk.kpad, cls_slot)
            tok2 = torch.empty((rows, D), dtype=torch.bfloat16, device=DEV)
            ops.gemm(A, pk, tok2, ops.EPI_BIAS, rows)
            got2 = tok2.view(B, Np + int(cls_slot), D)[:, int(cls_slot):]
            assert torch.equal(got, got2), "gathered and patch-matrix routes differ"


@pytest.mark.parametrize("D", [2056, 3072, 8192])
def test_row_kernels_wide_rows(ops, D):
    """LayerNorm statistics / LayerNorm / embed_finalize on rows wider than 2048 values (16 chunks per lane)."""
    M = 37
    F = torch.nn.functional
    x = _bf(_rand(M, D, seed=70) * 1.7 + 0.3)
    w, b = 1 + 0.1 * _rand(D, seed=71), 0.1 * _rand(D, seed=72)
    st = ops.ln_stats(x, M, D, 1e-5)
    xf = x.float()
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    _close(st[:, 0], mean, rel=1e-5, what="wide ln_stats mean")
    _close(st[:, 1], torch.rsqrt(var + 1e-5), rel=1e-5, what="wide ln_stats rstd")
    for dt, rel in ((torch.bfloat16, 6e-3), (torch.float32, 1e-5)):
        out = torch.empty((M, D), dtype=dt, device=DEV)
        ops.layernorm(x, out, M, D, w, b, 1e-5)
        _close(out, F.layer_norm(xf, (D,), w, b, 1e-5), rel=rel, what=f"wide layernorm {dt}")
    Bn, N = 2, 9
    tok = _bf(_rand(Bn * N, D, seed=73))
    cls, pos = _rand(D, seed=74), _rand(N, D, seed=75)
    y = tok.clone()
    ops.embed_finalize(y, Bn, N, D, cls, pos, None, None, 0.0, w, b, 1e-5)
    t = tok.float().view(Bn, N, D).clone()
    t[:, 0] = cls
    _close(y.view(Bn, N, D), F.layer_norm(t, (D,), w, b, 1e-5) + pos, rel=6e-3, what="wide embed_finalize")


@pytest.mark.parametrize("C,H,Wd,ph,pw,D,B", [(3, 96, 96, 32, 32, 256, 3700),    # K = 3072, 9 patches per image
                                              (1, 64, 128, 16, 16, 256, 1100),    # one channel, non-square image
                                              (3, 32, 64, 8, 16, 512, 2100),      # non-square patch: K = 384
                                              (4, 48, 48, 4, 8, 256, 1000),       # 64 / pw = 8 image rows per K-tile > ph
                                              (2, 64, 64, 2, 64, 256, 1100)])     # a K-tile is one patch row
def test_patch_embed_gathering_loader_shapes(ops, C, H, Wd, ph, pw, D, B):
    """The big-tile kernel's gathering loader against the patch-matrix route (same kernel fed from memory), bit for
    bit, on patch / image / channel shapes that stress its address arithmetic; and against the oracle's im2col."""
    from oracle import vit_oracle as O
    from vitm_amd import _lib
    x = _bf(_rand(B, C, H, Wd, seed=80))
    w = _rand(D, C, ph, pw, seed=81, scale=(C * ph * pw) ** -0.5)
    b = _rand(D, seed=82, scale=0.1)
    pk = ops.pack_linear(w, b)
    Np = (H // ph) * (Wd // pw)
    assert pk.kpad == C * ph * pw and _lib.load().vitm_gemm_uses_big_tile(B * Np, pk.npad, pk.kpad) == 1
    for cls_slot in (False, True):
        rows = B * (Np + int(cls_slot))
        tok = torch.full((rows, D), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.patch_embed(x, pk, tok, (ph, pw), D, cls_slot)
        got = tok.view(B, Np + int(cls_slot), D)
        if cls_slot:
            assert (got[:, 0] == 7.0).all(), "class-token slot must be left untouched"
            got = got[:, 1:]
        A = torch.empty((rows, pk.kpad), dtype=torch.bfloat16, device=DEV)
        ops.im2col(x, A, (ph, pw), pk.kpad, cls_slot)
        tok2 = torch.empty((rows, D), dtype=torch.bfloat16, device=DEV)
        ops.gemm(A, pk, tok2, ops.EPI_BIAS, rows)
        assert torch.equal(got, tok2.view(B, Np + int(cls_slot), D)[:, int(cls_slot):]), "gathered != patch-matrix route"
        cols = O.im2col(x[:8].float().cpu(), (ph, pw)).to(DEV)
        _close(got[:8], cols @ _bf(w).float().reshape(D, -1).t() + b, what="gathering loader vs oracle im2col")


def test_embed_finalize_and_layernorm(ops):
    B, N, D = 3, 65, 192
    tok = _bf(_rand(B * N, D, seed=25))
    cls, pos = _rand(D, seed=26), _rand(N, D, seed=27)
    ew, eb = 1 + 0.1 * _rand(D, seed=28), 0.1 * _rand(D, seed=29)
    nw, nb = 1 + 0.1 * _rand(D, seed=30), 0.1 * _rand(D, seed=31)
    x = tok.clone()
    ops.embed_finalize(x, B, N, D, cls, pos, ew, eb, 1e-5, nw, nb, 1e-5)
    F = torch.nn.functional
    t = tok.float().view(B, N, D).clone()
    t[:, 1:] = F.layer_norm(t[:, 1:], (D,), ew, eb, 1e-5)
    t[:, 0] = cls
    ref = F.layer_norm(t, (D,), nw, nb, 1e-5) + pos
    _close(x.view(B, N, D), ref, rel=6e-3, what="embed_finalize")
    # the same call also leaving the LayerNorm statistics of the finished rows
    x2 = tok.clone()
    st = torch.full((B * N, 2), float("nan"), device=DEV)
    ops.embed_finalize(x2, B, N, D, cls, pos, ew, eb, 1e-5, nw, nb, 1e-5, stats=st, stats_eps=1e-6)
    assert torch.equal(x2, x)
    st_ref = ops.ln_stats(x2, B * N, D, 1e-6)   # the statistics are those of the rows as stored
    assert torch.equal(st, st_ref)
    # pos only
    x = tok.clone()
    ops.embed_finalize(x, B, N, D, cls, pos, None, None, 0.0, None, None, 0.0)
    t = tok.float().view(B, N, D).clone()
    t[:, 0] = cls
    _close(x.view(B, N, D), t + pos, rel=6e-3, what="embed_finalize (no norms)")
    # full layernorm, bf16 and fp32 out
    for dt, rel in ((torch.bfloat16, 6e-3), (torch.float32, 1e-5)):
        out = torch.empty((B * N, D), dtype=dt, device=DEV)
        ops.layernorm(tok, out, B * N, D, nw, nb, 1e-5)
        _close(out, F.layer_norm(tok.float(), (D,), nw, nb, 1e-5), rel=rel, what=f"layernorm {dt}")


def test_scale_and_casts(ops):
    M, D = 50, 136
    x = _rand(M, D, seed=32)
    xb = ops.to_bf16(x)
    assert torch.equal(xb, x.to(torch.bfloat16))
    assert torch.equal(ops.to_f32(xb), xb.float())
    g = _rand(D, seed=33)
    out = torch.empty_like(xb)
    ops.scale_channels(xb, out, M, D, g)
    _close(out, xb.float() * g, rel=5e-3, what="scale")


def test_bad_arguments_report_errors(ops):
    from vitm_amd import VitmError
    A = torch.zeros((8, 100), dtype=torch.bfloat16, device=DEV)
    pk = ops.pack_linear(torch.zeros((8, 64), device=DEV), None)
    with pytest.raises((VitmError, AssertionError)):
        ops.gemm(A[:, :32], pk, torch.empty((8, 8), dtype=torch.bfloat16, device=DEV), ops.EPI_BIAS, 8)
    with pytest.raises(VitmError):
        ops.attention(torch.zeros((4, 3 * 136), dtype=torch.bfloat16, device=DEV),
                      torch.zeros((4, 136), dtype=torch.bfloat16, device=DEV), 1, 4, 1, 136, 0.1)
