"""fp32 training kernels on a real MI355X against float64 PyTorch (autograd where a gradient is checked).
Bar: 1e-4 relative (max|got - ref| / max|ref|), 1e-5 for the element-wise ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
F = torch.nn.functional


@pytest.fixture(scope="module")
def T():
    from vitm_amd import _train_ops
    return _train_ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _rel(got, ref):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,mp", [(100, 70, 128), (33, 257, 64), (5000, 192, 5056), (64, 64, 64)])
def test_transpose(T, M, N, mp):
    src = _rand(M, N + 8, seed=1)[:, :N]
    dst = T.transpose(src, M, N, mp)
    assert torch.equal(dst[:, :M], src.t()) and (dst[:, M:] == 0).all()


@pytest.mark.parametrize("M,D", [(1, 8), (50, 100), (5000, 768), (197 * 16, 614)])
def test_colsum(T, M, D):
    a, b = _rand(M, D, seed=2), _rand(M, D, seed=3)
    assert _rel(T.colsum(a, M, D), a.double().sum(0)) <= 1e-5
    assert _rel(T.colsum(a, M, D, b), (a.double() * b.double()).sum(0)) <= 1e-5
    assert torch.equal(T.colsum(a, M, D, b), T.colsum(a, M, D, b)), "deterministic"


@pytest.mark.parametrize("M,D", [(7, 64), (300, 192), (2000, 768), (65, 1024), (9001, 768), (13000, 1536)])   # the last two: several rows per wave
def test_layernorm_bwd(T, M, D):
    x = (_rand(M, D, seed=4) * 2 + 0.5).double().requires_grad_(True)
    w = (1 + 0.2 * _rand(D, seed=5)).double().requires_grad_(True)
    b = (0.1 * _rand(D, seed=6)).double().requires_grad_(True)
    dy, add = _rand(M, D, seed=7), _rand(M, D, seed=8)
    F.layer_norm(x, (D,), w, b, 1e-5).backward(dy.double())
    dx = torch.empty((M, D), device=DEV)
    dw, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    T.layernorm_bwd(x.detach().float(), dy, w.detach().float(), 1e-5, M, D, dx, dx_add=add, dw_out=dw, db_out=db)
    assert _rel(dx, x.grad + add.double()) <= 1e-4
    assert _rel(dw, w.grad) <= 1e-4 and _rel(db, b.grad) <= 1e-4
    T.layernorm_bwd(x.detach().float(), dy, w.detach().float(), 1e-5, M, D, dx)
    assert _rel(dx, x.grad) <= 1e-4


@pytest.mark.parametrize("M,N,K,with_bias", [(300, 96, 200, True), (513, 768, 3072, True), (64, 40, 72, False)])
def test_layerscale_grads_from_weight_gradient(T, M, N, K, with_bias):
    """y = gamma * (h W^T + b): dgamma, dW and db out of the UNSCALED weight / bias gradients equal float64 autograd."""
    h, dy = _rand(M, K, seed=51).double(), _rand(M, N, seed=52).double()
    w = (0.1 * _rand(N, K, seed=53)).double().requires_grad_(True)
    b = (0.1 * _rand(N, seed=54)).double().requires_grad_(True) if with_bias else None
    g = (0.5 + _rand(N, seed=55).abs()).double().requires_grad_(True)
    z = h @ w.t() + (b if with_bias else 0)
    (g * z).backward(dy)
    dw = (dy.t() @ h).float().contiguous()            # what the weight-gradient GEMM delivers without gamma
    db = dy.sum(0).float().contiguous() if with_bias else None
    dg = torch.empty(N, device=DEV)
    T.layerscale_grads(w.detach().float(), b.detach().float() if with_bias else None, dw, db, g.detach().float(), dg,
                       round_bf16=False)
    assert _rel(dg, g.grad) <= 1e-5 and _rel(dw, w.grad) <= 1e-5
    if with_bias:
        assert _rel(db, b.grad) <= 1e-5


def test_gelu_bwd(T):
    M, D = 300, 200
    pre = (_rand(M, D, seed=9) * 2).double().requires_grad_(True)
    dh = _rand(M, D, seed=10)
    F.gelu(pre).backward(dh.double())
    out = torch.empty((M, D), device=DEV)
    T.gelu_bwd(pre.detach().float(), dh, out, M, D)
    assert _rel(out, pre.grad) <= 1e-5


@pytest.mark.parametrize("B,C", [(4, 10), (64, 10), (256, 1000), (3, 7)])
def test_cross_entropy(T, B, C):
    logits = (_rand(B, C, seed=11) * 3).double().requires_grad_(True)
    labels = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(12)).to(DEV)
    ref = F.cross_entropy(logits, labels)
    ref.backward()
    loss, dl = T.cross_entropy(logits.detach().float(), labels)
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert _rel(dl, logits.grad) <= 1e-5


def test_grad_clip_matches_torch(T):
    for scale in (0.01, 5.0):
        g = _rand(1_234_567, seed=13, scale=scale)
        ref = g.clone().requires_grad_(False)
        p = torch.nn.Parameter(torch.zeros_like(ref))
        p.grad = ref
        norm = torch.nn.utils.clip_grad_norm_([p], 1.0)
        nc = T.grad_clip(g, 1.0)
        assert abs(nc[0].item() - norm.item()) <= 1e-5 * norm.item()
        assert _rel(g, p.grad) <= 1e-5


def test_adamw_matches_torch(T):
    n = 100_003
    p0, g1, g2 = _rand(n, seed=14), _rand(n, seed=15, scale=0.1), _rand(n, seed=16, scale=0.1)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step, g in enumerate((g1, g2), 1):
        ref.grad = g.clone()
        opt.step()
        T.adamw(p, g, m, v, 1e-3, 0.9, 0.98, 1e-8, 0.05, step)
        assert _rel(p, ref.detach()) <= 1e-6
    st = opt.state[ref]
    assert _rel(m, st["exp_avg"]) <= 1e-6 and _rel(v, st["exp_avg_sq"]) <= 1e-6


@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (197 * 3, 768, 192), (65, 10, 192), (1000, 614, 768)])
def test_linear_backward_from_forward_gemm(T, M, N, K):
    x = _rand(M, K, seed=17).double().requires_grad_(True)
    w = (_rand(N, K, seed=18) * K ** -0.5).double().requires_grad_(True)
    dy = _rand(M, N, seed=19)
    (x @ w.t()).backward(dy.double())
    dx = T.linear_dx(dy, w.detach().float(), M)
    assert _rel(dx[:, :K], x.grad) <= 1e-4
    dw = T.linear_dw(dy, x.detach().float(), M, N, K)
    assert _rel(dw, w.grad) <= 1e-4


@pytest.mark.parametrize("B,N,H,hd", [(2, 65, 3, 64), (1, 197, 2, 64), (2, 50, 2, 96), (1, 33, 2, 32), (1, 100, 1, 128),
                                      (1, 300, 1, 64), (2, 1, 2, 64)])
def test_attention_backward(T, B, N, H, hd):
    from vitm_amd import _ops
    D = H * hd
    qkv = (_rand(B * N, 3 * D, seed=20) * 1.2)
    d_o = _rand(B * N, D, seed=21)
    q64 = qkv.double().requires_grad_(True)
    q, k, v = q64.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref_o = (((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    ref_o.backward(d_o.double())
    o = torch.empty((B * N, D), device=DEV)
    lse = torch.empty(B * H * N, device=DEV)
    T.attention_lse(qkv, o, lse, B, N, H, hd, hd ** -0.5)
    o2 = torch.empty_like(o)
    _ops.attention(qkv, o2, B, N, H, hd, hd ** -0.5)
    assert torch.equal(o, o2), "the lse variant computes the same output"
    assert _rel(o, ref_o.detach()) <= 1e-4
    dqkv = torch.full((B * N, 3 * D), float("nan"), device=DEV)
    T.attention_bwd(qkv, o, d_o, lse, dqkv, B, N, H, hd, hd ** -0.5)
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel(dqkv[:, sl], q64.grad[:, sl]) <= 1e-4, name


# ---- bf16 variants (mixed-precision training): inputs are bf16-exact, so the only error is the rounding of the stored
# ---- result (2^-9 relative) plus fp32 summation; bars are 1e-2 of the largest entry for bf16 outputs, 1e-4 for fp32 ones
BF = torch.bfloat16


def test_transpose_bf16(T):
    M, N, mp = 333, 200, 384
    src = _rand(M, N + 8, seed=31).to(BF)[:, :N]
    dst = T.transpose(src, M, N, mp)
    assert dst.dtype == BF and torch.equal(dst[:, :M], src.t()) and (dst[:, M:] == 0).all()


def test_colsum_bf16(T):
    M, D = 3000, 200
    a, b = _rand(M, D, seed=32).to(BF), _rand(M, D, seed=33).to(BF)
    got = T.colsum(a, M, D, b)
    assert got.dtype == torch.float32
    assert _rel(got, (a.double() * b.double()).sum(0)) <= 1e-5
    assert _rel(T.colsum(a, M, D), a.double().sum(0)) <= 1e-5


@pytest.mark.parametrize("M,D", [(500, 192), (20011, 384), (50432, 768)])
def test_layernorm_bwd_bf16(T, M, D):
    x = (_rand(M, D, seed=34) * 2 + 0.5).to(BF)
    dy, add = _rand(M, D, seed=35).to(BF), _rand(M, D, seed=36).to(BF)
    w = (1 + 0.2 * _rand(D, seed=37))
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = torch.zeros(D, device=DEV, dtype=torch.float64, requires_grad=True)
    F.layer_norm(x64, (D,), w64, b64, 1e-5).backward(dy.double())
    dx = torch.empty((M, D), device=DEV, dtype=BF)
    dw, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    T.layernorm_bwd(x, dy, w, 1e-5, M, D, dx, dx_add=add, dw_out=dw, db_out=db)
    assert _rel(dx, x64.grad + add.double()) <= 1e-2
    assert _rel(dw, w64.grad) <= 1e-4 and _rel(db, b64.grad) <= 1e-4
    dx2, dw2, db2 = torch.empty_like(dx), torch.empty_like(dw), torch.empty_like(db)
    T.layernorm_bwd(x, dy, w, 1e-5, M, D, dx2, dx_add=add, dw_out=dw2, db_out=db2)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2), "deterministic"


def test_gelu_bwd_and_residual_scale_bf16(T):
    M, D = 300, 200
    pre = (_rand(M, D, seed=38) * 2).to(BF)
    dh = _rand(M, D, seed=39).to(BF)
    p64 = pre.double().requires_grad_(True)
    F.gelu(p64).backward(dh.double())
    out = torch.empty((M, D), device=DEV, dtype=BF)
    T.gelu_bwd(pre, dh, out, M, D)
    assert _rel(out, p64.grad) <= 1e-2
    g = 1 + 0.1 * _rand(D, seed=40)
    T.residual_scale(pre, dh, g, out, M, D)
    assert _rel(out, pre.double() + g.double() * dh.double()) <= 1e-2


@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (197 * 3, 768, 192), (65, 10, 192)])
def test_linear_backward_bf16(T, M, N, K):
    x = _rand(M, K, seed=41).to(BF)
    w = (_rand(N, K, seed=42) * K ** -0.5).to(BF).float()          # fp32 master weight holding bf16-exact values
    dy = _rand(M, N, seed=43).to(BF)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (x64 @ w64.t()).backward(dy.double())
    dx = T.linear_dx(dy, w, M)
    assert dx.dtype == BF and _rel(dx[:, :K], x64.grad) <= 1e-2
    dw = T.linear_dw(dy, x, M, N, K)
    assert dw.dtype == torch.float32 and _rel(dw, w64.grad) <= 1e-4   # fp32 accumulators stored as fp32


@pytest.mark.parametrize("B,N,H,hd", [(2, 65, 3, 64), (1, 197, 2, 64), (2, 50, 2, 96), (1, 33, 2, 32), (1, 100, 1, 128),
                                      (1, 300, 1, 64), (2, 1, 2, 64),
                                      # the one-workgroup kernel at its limits: 7 and 8 whole tiles, one row into the eighth
                                      (1, 224, 2, 32), (1, 225, 1, 64), (1, 256, 2, 64), (1, 257, 1, 64)])
def test_attention_backward_bf16(T, B, N, H, hd):
    from vitm_amd import _ops
    D = H * hd
    qkv = (_rand(B * N, 3 * D, seed=44) * 1.2).to(BF)
    d_o = _rand(B * N, D, seed=45).to(BF)
    q64 = qkv.double().requires_grad_(True)
    q, k, v = q64.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q * hd ** -0.5) @ k.transpose(-2, -1)
    ref_o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    ref_o.backward(d_o.double())
    o = torch.empty((B * N, D), device=DEV, dtype=BF)
    lse = torch.full((B * H * N,), float("nan"), device=DEV)
    T.attention_lse(qkv, o, lse, B, N, H, hd, hd ** -0.5)
    o2 = torch.empty_like(o)
    _ops.attention(qkv, o2, B, N, H, hd, hd ** -0.5)
    assert torch.equal(o, o2), "the lse variant computes the same output"
    ref_lse = (torch.logsumexp(s.detach(), -1) * 1.4426950408889634).reshape(-1)      # log2 domain, (b, h, n)
    assert (lse.double() - ref_lse).abs().max().item() <= 1e-3
    dqkv = torch.full((B * N, 3 * D), float("nan"), device=DEV, dtype=BF)
    T.attention_bwd(qkv, o, d_o, lse, dqkv, B, N, H, hd, hd ** -0.5)
    floor = 1e-3 * q64.grad.abs().max().item()          # one key: dq and dk are exactly zero in the reference
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        ref = q64.grad[:, sl]
        err = (dqkv[:, sl].double() - ref).abs().max().item() / max(ref.abs().max().item(), floor)
        assert torch.isfinite(dqkv[:, sl]).all() and err <= 2e-2, (name, err)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1, 8, 8), (700, 200, 136), (12608, 768, 768), (5000, 10, 192),
                                   (3000, 616, 192), (197 * 64, 2304, 768)])
def test_gemm_tn(T, M, N, K):
    """dW = a^T b straight from the row-major activations; garbage in the pad columns of a row must not leak."""
    lda, ldb = (N + 7) // 8 * 8 + 8, (K + 7) // 8 * 8 + 16
    a = torch.full((M, lda), float("nan"), device=DEV, dtype=BF)
    b = torch.full((M, ldb), float("nan"), device=DEV, dtype=BF)
    a[:, :N] = _rand(M, N, seed=51).to(BF)
    b[:, :K] = _rand(M, K, seed=52).to(BF)
    nr = (N + 7) // 8 * 8
    kr = (K + 7) // 8 * 8
    a[:, N:nr] = 0          # the 16-byte chunk that straddles the last column is read: it must hold numbers
    b[:, K:kr] = 0
    cs = torch.full((N,), float("nan"), device=DEV)
    out = T.gemm_tn(a, b, M, N, K, colsum_out=cs)
    ref = a[:, :N].double().t() @ b[:, :K].double()
    assert _rel(out, ref) <= 1e-5
    assert _rel(cs, a[:, :N].double().sum(0)) <= 1e-5, "bias gradient from the same pass"
    assert torch.equal(out, T.gemm_tn(a, b, M, N, K)), "deterministic"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H,hd", [(70, 3, 64), (33, 2, 32), (50, 2, 96), (20, 1, 128), (1, 1, 64)])
def test_qk_norm_backward(T, dt, M, H, hd):
    D = H * hd
    qkv = (_rand(M, 3 * D, seed=61) * 1.5 + 0.3).to(dt)
    dy = _rand(M, 3 * D + 8, seed=62).to(dt)[:, :3 * D]            # a strided view, like the parallel block's buffer
    qw, qb = 1 + 0.2 * _rand(hd, seed=63), 0.1 * _rand(hd, seed=64)
    kw, kb = 1 + 0.2 * _rand(hd, seed=65), 0.1 * _rand(hd, seed=66)
    x64 = qkv.double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (qw, qb, kw, kb)]
    q, k, v = x64.reshape(M, 3, H, hd).unbind(1)
    out = torch.stack((F.layer_norm(q, (hd,), p64[0], p64[1], 1e-5), F.layer_norm(k, (hd,), p64[2], p64[3], 1e-6), v), 1)
    out.reshape(M, 3 * D).backward(dy.double())
    d = dy.clone() if False else dy        # in place on the view
    base = d.clone()
    outs = [torch.full((hd,), float("nan"), device=DEV) for _ in range(4)]
    T.qk_norm_bwd(qkv, d, M, H, hd, qw, 1e-5, kw, 1e-6, *outs)
    tol = 1e-4 if dt == torch.float32 else 2e-2
    assert _rel(d[:, :2 * D], x64.grad[:, :2 * D]) <= tol
    assert torch.equal(d[:, 2 * D:], base[:, 2 * D:]), "v gradients untouched"
    for got, ref in zip(outs, p64):
        assert _rel(got, ref.grad) <= 1e-4


@pytest.mark.parametrize("M,N,K", [(200, 136, 128), (65, 192, 614), (12608, 768, 3072), (197 * 8, 768, 3072)])
def test_linear_dx_with_fused_gelu_backward(T, M, N, K):
    """dpre = (dy . W) * GELU'(pre) in the GEMM epilogue (both the 128x128 and the persistent 256x256 kernel)."""
    from vitm_amd import _ops
    dy = _rand(M, N, seed=71).to(BF)
    w = (_rand(N, K, seed=72) * N ** -0.5).to(BF).float()
    kp = _ops.round_up(K, 64)
    pre = torch.zeros((M, kp), device=DEV, dtype=BF)
    pre[:, :K] = (_rand(M, K, seed=73) * 2).to(BF)
    p64 = pre[:, :K].double().requires_grad_(True)
    F.gelu(p64).backward(dy.double() @ w.double())
    got = T.linear_dx(dy, w, M, gelu_input=pre)
    assert got.dtype == BF and _rel(got[:, :K], p64.grad) <= 2e-2
    # against the unfused pair of kernels, to bf16 rounding of the intermediate
    dh = T.linear_dx(dy, w, M)
    ref = torch.empty_like(dh)
    T.gelu_bwd(pre, dh, ref, M, K)
    assert _rel(got[:, :K], ref[:, :K].double()) <= 2e-2


def test_layerscale_folded_into_linear_backward(T):
    """dy * gamma never materialised: gamma scales W^T's source rows (input gradient) and the rows of dW / db."""
    M, N, K = 1500, 768, 256
    dy = _rand(M, N, seed=81).to(BF)
    x = _rand(M, K, seed=82).to(BF)
    w = (_rand(N, K, seed=83) * N ** -0.5).to(BF).float()
    gam = 1 + 0.3 * _rand(N, seed=84)
    dz = dy.double() * gam.double()
    db = torch.full((N,), float("nan"), device=DEV)
    dw = T.linear_dw(dy, x, M, N, K, bias_out=db, dy_scale=gam)
    assert _rel(dw, dz.t() @ x.double()) <= 1e-5 and _rel(db, dz.sum(0)) <= 1e-5
    dx = T.linear_dx(dy, w, M, dy_scale=gam)
    assert _rel(dx[:, :K], dz @ w.double()) <= 2e-2
    wt = T.transpose(w, N, K, 832, out_dtype=BF, row_scale=gam)
    assert _rel(wt[:, :N], (w.double() * gam.double()[:, None]).t()) <= 1e-2 and (wt[:, N:] == 0).all()
