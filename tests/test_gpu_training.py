"""The training step on a real MI355X against the fixture captured from the imported reference
(tests/golden/make_golden_train.py: three iterations of the loop body of vitm/training/train.py:215-224 on the
``tiny2`` configuration, fp32 CPU).  Bars: loss and gradient norm 1e-4 relative; every parameter gradient of step 1
within 1e-4 of the largest gradient entry of that tensor (2e-4 for the deepest ones); parameters after three AdamW
steps within 2e-5 absolute (the update is lr-sized: 1e-5 .. 1.4e-5)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import weights as W

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(cfg, seed):
    from vitm_amd.models import VisionTransformer, VITC
    LN = nn.LayerNorm
    vt = VisionTransformer(cfg["image_size"], patch_size=cfg["patch_size"], num_classes=cfg["num_classes"],
                           num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"], depth=cfg["depth"],
                           embed_dim=cfg["embed_dim"], norm_layer=LN, embed_norm_layer=LN if cfg["embed_norm"] else None,
                           final_norm_layer=LN if cfg["final_norm"] else None)
    m = VITC(vt).train()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, seed).items()}, strict=True)
    return m.cuda()


def test_training_steps_match_reference():
    from vitm_amd import training
    g = np.load(os.path.join(GOLD, "train_tiny2.npz"))
    cfg = W.CONFIGS["tiny2"]
    seed, batch, steps = int(g["seed"]), int(g["batch"]), int(g["steps"])
    model = _model(cfg, seed)
    opt, sched = training.reference_optimizer_and_scheduler(model, 1000)
    tr = training.Trainer(model, opt, sched, max_norm=1.0)
    for step in range(steps):
        x = torch.from_numpy(W.make_input(cfg, batch, seed + 10 + step)).cuda()
        y = torch.from_numpy(g[f"labels{step}"]).cuda()
        if step == 0:   # raw gradients of the first step, through the plain autograd interface
            tr.grads.zero()
            logits = model(x)
            assert logits.requires_grad and logits.shape == (batch, cfg["num_classes"])
            loss = training.CrossEntropyLoss()(logits, y)
            loss.backward()
            assert abs(loss.item() - g["losses"][0]) <= 1e-4 * abs(g["losses"][0])
            worst = 0.0
            for k, p in model.named_parameters():
                ref = torch.from_numpy(g["grad." + k])
                assert p.grad is not None and p.grad.shape == ref.shape, k
                err = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
                worst = max(worst, err)
                assert err <= 2e-4, f"{k}: gradient relative error {err:.3g}"
            print("worst relative gradient error", worst)
        assert abs(opt.param_groups[0]["lr"] - g["lrs"][step]) <= 1e-12
        loss = tr.step(x, y)
        assert abs(loss.item() - g["losses"][step]) <= 1e-4 * abs(g["losses"][step]), (step, loss.item())
        assert abs(tr.last_grad_norm.item() - g["grad_norms"][step]) <= 2e-4 * g["grad_norms"][step], step
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g["param." + k])
        assert (p.detach().cpu() - ref).abs().max().item() <= 2e-5, k
    # the optimizer state has torch's layout
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == steps


def test_training_without_optional_norms_and_with_padding(monkeypatch):
    """No embed / pre / final norm, head_dim 96 with a hidden width (614) that needs padding, against float64 autograd
    of the CPU oracle."""
    from oracle import vit_oracle as O
    from vitm_amd import training
    from vitm_amd.models import VisionTransformer, VITC
    vt = VisionTransformer(32, patch_size=4, num_classes=10, num_heads=2, mlp_ratio=0.8, depth=2, embed_dim=192)
    model = VITC(vt).train()
    torch.manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.normal_(0, 0.1)
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x = torch.randn(5, 3, 32, 32)
    y = torch.tensor([1, 0, 9, 3, 3])
    ocfg = dict(image_size=(32, 32), patch_size=(4, 4), num_heads=2, depth=2)
    # the oracle reads its weights detached; for this test it must keep the autograd graph
    monkeypatch.setattr(O, "_t", lambda sd, key, dtype: None if sd.get(key) is None else sd[key].to(dtype))
    logits64 = O.vitc_forward(x.double(), sd64, ocfg, dtype=torch.float64)
    loss64 = torch.nn.functional.cross_entropy(logits64, y)
    loss64.backward()
    model = model.cuda()
    logits = model(x.cuda())
    loss = training.CrossEntropyLoss()(logits, y.cuda())
    loss.backward()
    assert abs(loss.item() - loss64.item()) <= 1e-4 * abs(loss64.item())
    for k, p in model.named_parameters():
        ref = sd64[k].grad
        err = (p.grad.cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
        assert err <= 2e-4, f"{k}: {err:.3g}"


def test_inference_path_unchanged_under_no_grad():
    model = _model(W.CONFIGS["tiny2"], 1)
    x = torch.from_numpy(W.make_input(W.CONFIGS["tiny2"], 3, 2)).cuda()
    with torch.no_grad():
        a = model(x)
    b = model(x)
    assert not a.requires_grad and b.requires_grad
    assert (a - b.detach()).abs().max().item() <= 1e-4 * a.abs().max().item()


# ---- bf16 mixed precision -----------------------------------------------------------------------------------------
# The reference has no bf16 training run to compare with (train.py has no autocast), so the bars are those of a
# mixed-precision step against the reference's fp32 one: loss within 1e-2 relative, gradient norm within 3e-2, every
# parameter gradient at cosine >= 0.99 to the reference's (0.95 for tensors of fewer than 1000 entries, where a few
# rounded entries move the angle), parameters after three lr-sized AdamW steps within 1e-4 absolute and 1e-5 on average.
def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def test_bf16_training_steps_track_reference():
    from vitm_amd import training
    g = np.load(os.path.join(GOLD, "train_tiny2.npz"))
    cfg = W.CONFIGS["tiny2"]
    seed, batch, steps = int(g["seed"]), int(g["batch"]), int(g["steps"])
    model = _model(cfg, seed)
    opt, sched = training.reference_optimizer_and_scheduler(model, 1000)
    tr = training.Trainer(model, opt, sched, max_norm=1.0, compute_dtype=torch.bfloat16)
    for step in range(steps):
        x = torch.from_numpy(W.make_input(cfg, batch, seed + 10 + step)).cuda()
        y = torch.from_numpy(g[f"labels{step}"]).cuda()
        if step == 0:
            tr.grads.zero()
            logits = model(x.to(torch.bfloat16))         # bf16 images select the mixed-precision path
            assert logits.requires_grad and logits.dtype == torch.float32
            loss = training.CrossEntropyLoss()(logits, y)
            loss.backward()
            assert abs(loss.item() - g["losses"][0]) <= 1e-2 * abs(g["losses"][0])
            worst = 1.0
            for k, p in model.named_parameters():
                ref = torch.from_numpy(g["grad." + k])
                assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
                c = _cos(p.grad.cpu(), ref)
                worst = min(worst, c)
                assert c >= (0.99 if ref.numel() >= 1000 else 0.95), f"{k}: gradient cosine {c:.4f}"
            print("worst gradient cosine", worst)
        loss = tr.step(x, y)                              # fp32 images, compute_dtype=bf16
        assert abs(loss.item() - g["losses"][step]) <= 1e-2 * abs(g["losses"][step]), (step, loss.item())
        assert abs(tr.last_grad_norm.item() - g["grad_norms"][step]) <= 3e-2 * g["grad_norms"][step], step
    for k, p in model.named_parameters():
        assert p.dtype == torch.float32
        d = (p.detach().cpu() - torch.from_numpy(g["param." + k])).abs()
        assert d.max().item() <= 1e-4 and d.mean().item() <= 1e-5, (k, d.max().item(), d.mean().item())


def test_bf16_training_with_padding_tracks_float64(monkeypatch):
    from oracle import vit_oracle as O
    from vitm_amd import training
    from vitm_amd.models import VisionTransformer, VITC
    vt = VisionTransformer(32, patch_size=4, num_classes=10, num_heads=2, mlp_ratio=0.8, depth=2, embed_dim=192)
    model = VITC(vt).train()
    torch.manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.normal_(0, 0.1)
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x = torch.randn(5, 3, 32, 32).to(torch.bfloat16)
    y = torch.tensor([1, 0, 9, 3, 3])
    ocfg = dict(image_size=(32, 32), patch_size=(4, 4), num_heads=2, depth=2)
    monkeypatch.setattr(O, "_t", lambda sd, key, dtype: None if sd.get(key) is None else sd[key].to(dtype))
    loss64 = torch.nn.functional.cross_entropy(O.vitc_forward(x.double(), sd64, ocfg, dtype=torch.float64), y)
    loss64.backward()
    model = model.cuda()
    loss = training.CrossEntropyLoss()(model(x.cuda()), y.cuda())
    loss.backward()
    assert abs(loss.item() - loss64.item()) <= 1e-2 * abs(loss64.item())
    for k, p in model.named_parameters():
        c = _cos(p.grad.cpu(), sd64[k].grad)
        assert c >= (0.99 if p.numel() >= 1000 else 0.95), f"{k}: {c:.4f}"


def test_vit_l16_bf16_training_step_tracks_float64(monkeypatch):
    """ViT-L/16 shapes (embed 1024, 16 heads of 64, hidden 4096, 197 tokens, 1000 classes) at depth 2, batch 4:
    the bf16 mixed-precision step (autograd interface for the gradients, then Trainer.step) against float64 autograd of
    the oracle on the same weights; bars of test_bf16_training_with_padding_tracks_float64."""
    from oracle import vit_oracle as O
    from vitm_amd import training
    cfg = dict(W.CONFIGS["vit_l16"], depth=2)
    model = _model(cfg, 31).cpu()
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x = torch.from_numpy(W.make_input(cfg, 4, 32)).to(torch.bfloat16)
    y = torch.tensor([5, 999, 0, 417])
    monkeypatch.setattr(O, "_t", lambda sd, key, dtype: None if sd.get(key) is None else sd[key].to(dtype))
    loss64 = torch.nn.functional.cross_entropy(O.vitc_forward(x.double(), sd64, W.oracle_cfg(cfg), dtype=torch.float64), y)
    loss64.backward()
    norm64 = torch.sqrt(sum((v.grad ** 2).sum() for v in sd64.values() if v.grad is not None)).item()
    model = model.cuda()
    loss = training.CrossEntropyLoss()(model(x.cuda()), y.cuda())
    loss.backward()
    assert abs(loss.item() - loss64.item()) <= 1e-2 * abs(loss64.item())
    for k, p in model.named_parameters():
        c = _cos(p.grad.cpu(), sd64[k].grad)
        assert c >= (0.99 if p.numel() >= 1000 else 0.95), f"{k}: {c:.4f}"
    # the whole step: forward, loss, backward into the flat buffer, clip to 1.0, AdamW
    opt, sched = training.reference_optimizer_and_scheduler(model, 1000)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    tr = training.Trainer(model, opt, sched, max_norm=1.0, compute_dtype=torch.bfloat16)
    step_loss = tr.step(x.cuda(), y.cuda())
    assert abs(step_loss.item() - loss64.item()) <= 1e-2 * abs(loss64.item())
    assert abs(tr.last_grad_norm.item() - norm64) <= 3e-2 * norm64
    moved = max((p.detach() - before[k]).abs().max().item() for k, p in model.named_parameters())
    assert 0 < moved <= 2e-5      # first step: lr = 0.01 * 1e-3, AdamW update of unit size


def test_vit_l16_bf16_training_step_full_size_route(monkeypatch):
    """Benchmark configuration 5 at the size it is benchmarked: ViT-L/16 shapes, batch 128 (M = 25 216 rows: the
    256x256-tile GEMMs with carried moments in the forward, the multi-slice weight-gradient GEMM, the persistent
    attention kernels), depth 2.  The batch is the 4-image float64 case of the test above repeated 32 times: in exact
    arithmetic the mean loss and every mean gradient equal those of the 4 images, so the float64 autograd of the oracle
    on 4 images is the reference for the whole batch, at the bf16 bars of this file.  Two steps from one state are
    bit-identical (every reduction has a fixed order)."""
    import copy
    from oracle import vit_oracle as O
    from vitm_amd import training
    cfg = dict(W.CONFIGS["vit_l16"], depth=2)
    model = _model(cfg, 31).cpu()
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x4 = torch.from_numpy(W.make_input(cfg, 4, 32)).to(torch.bfloat16)
    y4 = torch.tensor([5, 999, 0, 417])
    monkeypatch.setattr(O, "_t", lambda sd, key, dtype: None if sd.get(key) is None else sd[key].to(dtype))
    loss64 = torch.nn.functional.cross_entropy(O.vitc_forward(x4.double(), sd64, W.oracle_cfg(cfg), dtype=torch.float64), y4)
    loss64.backward()
    norm64 = torch.sqrt(sum((v.grad ** 2).sum() for v in sd64.values() if v.grad is not None)).item()
    x, y = x4.repeat(32, 1, 1, 1).cuda(), y4.repeat(32).cuda()
    model = model.cuda()
    state = copy.deepcopy(model.state_dict())
    results = []
    for _ in range(2):
        model.load_state_dict(state)
        opt, sched = training.reference_optimizer_and_scheduler(model, 1000)
        tr = training.Trainer(model, opt, sched, max_norm=1.0, compute_dtype=torch.bfloat16)
        loss = tr.step(x, y)
        results.append((loss.clone(), tr.last_grad_norm.clone(), {k: p.detach().clone() for k, p in model.named_parameters()}))
    (l0, n0, p0), (l1, n1, p1) = results
    assert torch.equal(l0, l1) and torch.equal(n0, n1), "two steps from one state: loss / gradient norm differ"
    for k in p0:
        assert torch.equal(p0[k], p1[k]), f"{k}: parameters after the step differ between two runs"
    assert abs(l0.item() - loss64.item()) <= 1e-2 * abs(loss64.item()), (l0.item(), loss64.item())
    assert abs(n0.item() - norm64) <= 3e-2 * norm64, (n0.item(), norm64)
    # the gradients themselves, through the autograd interface on the same batch
    model.load_state_dict(state)
    tr.grads.zero()
    loss = training.CrossEntropyLoss()(model(x), y)
    loss.backward()
    for k, p in model.named_parameters():
        c = _cos(p.grad.cpu(), sd64[k].grad)
        assert c >= (0.99 if p.numel() >= 1000 else 0.95), f"{k}: {c:.4f}"


# ---- flat parameters / single-launch AdamW -------------------------------------------------------------------------
def _batch(cfg, batch, seed):
    x = torch.from_numpy(W.make_input(cfg, batch, seed)).cuda()
    y = torch.randint(0, cfg["num_classes"], (batch,), generator=torch.Generator().manual_seed(seed)).cuda()
    return x, y


def test_trainer_matches_autograd_loop_with_torch_adamw():
    """Trainer.step (gradients written in place, one AdamW launch over the flat buffers) against the reference's own
    loop body run through this package's autograd path with torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW."""
    from vitm_amd import training
    cfg = W.CONFIGS["tiny2"]
    a, b = _model(cfg, 5), _model(cfg, 5)
    opt_a = training.AdamW(a.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    tr = training.Trainer(a, opt_a, None, max_norm=1.0)
    assert tr.params is not None and tr.params.intact(), "parameters live in one flat buffer"
    opt_b = torch.optim.AdamW(b.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    crit = training.CrossEntropyLoss()
    for step in range(3):
        x, y = _batch(cfg, 6, 100 + step)
        la = tr.step(x, y)
        opt_b.zero_grad()
        lb = crit(b(x), y)
        lb.backward()
        nb = torch.nn.utils.clip_grad_norm_(b.parameters(), max_norm=1.0)
        opt_b.step()
        assert abs(la.item() - lb.item()) <= 1e-5 * abs(lb.item()), step
        assert abs(tr.last_grad_norm.item() - nb.item()) <= 1e-4 * nb.item(), step
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa - pb).abs().max().item() <= 2e-6, k
    sd = opt_a.state_dict()
    assert len(sd["state"]) == len(list(a.parameters())) and float(sd["state"][0]["step"]) == 3


def test_flat_adamw_state_reloads_and_detached_parameters_fall_back():
    import copy
    from vitm_amd import training
    cfg = W.CONFIGS["tiny2"]
    a = _model(cfg, 7)
    opt_a = training.AdamW(a.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    tr_a = training.Trainer(a, opt_a, None, max_norm=1.0, compute_dtype=torch.bfloat16)
    for step in range(2):
        tr_a.step(*_batch(cfg, 6, 200 + step))
    model_sd = {k: v.clone() for k, v in a.state_dict().items()}
    opt_sd = copy.deepcopy(opt_a.state_dict())
    x3, y3 = _batch(cfg, 6, 202)
    tr_a.step(x3, y3)
    # resume in a fresh trainer: bit-identical third step
    b = _model(cfg, 1)
    b.load_state_dict(model_sd)
    opt_b = training.AdamW(b.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    tr_b = training.Trainer(b, opt_b, None, max_norm=1.0, compute_dtype=torch.bfloat16)
    opt_b.load_state_dict(opt_sd)
    tr_b.step(x3, y3)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb), k
    # parameters re-allocated behind the trainer's back: the optimizer notices and updates tensor by tensor
    c = _model(cfg, 1)
    c.load_state_dict(model_sd)
    opt_c = training.AdamW(c.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    tr_c = training.Trainer(c, opt_c, None, max_norm=1.0, compute_dtype=torch.bfloat16)
    opt_c.load_state_dict(opt_sd)
    for p in c.parameters():
        p.data = p.data.clone()
    tr_c.step(x3, y3)
    assert opt_c._flat is None
    for (k, pa), (_, pc) in zip(a.named_parameters(), c.named_parameters()):
        assert torch.equal(pa, pc), k


# ---- attention variants and the parallel block: gradients against float64 autograd of the CPU oracle ------------------
def _randomise(model, seed):
    torch.manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma") or "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif n.endswith("bias"):
                p.normal_(0, 0.1)


def _check_grads(model, sd64, loss, loss64, bf16):
    if bf16:
        assert abs(loss.item() - loss64.item()) <= 1e-2 * abs(loss64.item())
    else:
        assert abs(loss.item() - loss64.item()) <= 1e-4 * abs(loss64.item())
    gmax = max(sd64[k].grad.abs().max().item() for k, _ in model.named_parameters())
    for k, p in model.named_parameters():
        ref = sd64[k].grad
        assert ref is not None and p.grad is not None and torch.isfinite(p.grad).all(), k
        if ref.abs().max().item() < 1e-9 * gmax:
            # mathematically zero (k_norm.bias: a shift common to every key leaves the softmax unchanged)
            assert p.grad.abs().max().item() <= (1e-3 if bf16 else 1e-5) * gmax, k
            continue
        if bf16:
            c = _cos(p.grad.cpu(), ref)
            assert c >= (0.99 if p.numel() >= 1000 else 0.95), f"{k}: cosine {c:.4f}"
        else:
            err = (p.grad.cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert err <= 2e-4, f"{k}: {err:.3g}"


@pytest.mark.parametrize("bf16", [False, True])
def test_training_with_qk_norm_and_qkv_bias(monkeypatch, bf16):
    """Sequential blocks whose attention has qkv_bias=True and qk_norm=True (models.py:54-55,82): backward through the
    per-head LayerNorm of q and k."""
    from oracle import vit_oracle as O
    from vitm_amd import training
    from vitm_amd.models import Attention, VisionTransformer, VITC
    LN = nn.LayerNorm
    vt = VisionTransformer(32, patch_size=8, num_classes=10, num_heads=2, mlp_ratio=2.0, depth=2, embed_dim=128,
                           norm_layer=LN, final_norm_layer=LN)
    for blk in vt.blocks:
        blk.attn = Attention(128, num_heads=2, qkv_bias=True, qk_norm=True)
    model = VITC(vt).train()
    _randomise(model, 11)
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x = torch.randn(4, 3, 32, 32)
    x = x.to(torch.bfloat16) if bf16 else x
    y = torch.tensor([1, 0, 9, 3])
    monkeypatch.setattr(O, "_t", lambda sd, key, dtype: None if sd.get(key) is None else sd[key].to(dtype))
    ocfg = dict(image_size=(32, 32), patch_size=(8, 8), num_heads=2, depth=2)
    loss64 = torch.nn.functional.cross_entropy(O.vitc_forward(x.double(), sd64, ocfg, dtype=torch.float64), y)
    loss64.backward()
    assert sd64["vit_model.blocks.0.attn.q_norm.weight"].grad is not None
    model = model.cuda()
    loss = training.CrossEntropyLoss()(model(x.cuda()), y.cuda())
    loss.backward()
    _check_grads(model, sd64, loss, loss64, bf16)


@pytest.mark.parametrize("bf16", [False, True])
def test_training_parallel_blocks(monkeypatch, bf16):
    """VisionTransformer(parallel_block=True): one step through Trainer-free autograd against float64 autograd of the
    oracle's parallel_block (parallel_block.py:193-221); hidden width 89 (not a multiple of 8) exercises the padding."""
    from oracle import vit_oracle as O
    from vitm_amd import training
    from vitm_amd.models import VisionTransformer, VITC
    LN = nn.LayerNorm
    vt = VisionTransformer(32, patch_size=8, num_classes=7, num_heads=2, mlp_ratio=0.7, depth=2, embed_dim=128,
                           norm_layer=LN, final_norm_layer=LN, parallel_block=True)
    model = VITC(vt).train()
    _randomise(model, 12)
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x = torch.randn(3, 3, 32, 32)
    x = x.to(torch.bfloat16) if bf16 else x
    y = torch.tensor([1, 0, 6])
    monkeypatch.setattr(O, "_t", lambda sd, key, dtype: None if sd.get(key) is None else sd[key].to(dtype))
    cfg = dict(image_size=32, patch_size=(8, 8), num_heads=2)
    f = O.forward_features(x.double(), sd64, cfg, "vit_model.", dtype=torch.float64)
    for i in range(2):
        f = O.parallel_block(f, sd64, 2, f"vit_model.blocks.{i}.", dtype=torch.float64)
    emb = O.layer_norm(f, sd64["vit_model.norm_final.weight"], sd64["vit_model.norm_final.bias"])
    loss64 = torch.nn.functional.cross_entropy(emb[:, 0] @ sd64["head.weight"].t() + sd64["head.bias"], y)
    loss64.backward()
    model = model.cuda()
    assert training.supported(model) is None
    loss = training.CrossEntropyLoss()(model(x.cuda()), y.cuda())
    loss.backward()
    _check_grads(model, sd64, loss, loss64, bf16)
    # and a Trainer step on the same model (gradients in place, flat AdamW) runs and lowers the loss on this batch
    opt = training.AdamW(model.parameters(), lr=1e-3)
    tr = training.Trainer(model, opt, None, max_norm=1.0)
    l0 = tr.step(x.cuda(), y.cuda()).item()
    for _ in range(3):
        l1 = tr.step(x.cuda(), y.cuda()).item()
    assert l1 < l0


def test_checkpoint_round_trip_of_a_trainer(tmp_path):
    """The end-of-training dictionary (train.py:245-249) written from a Trainer whose parameters and AdamW moments
    live in flat buffers: loads with weights_only=True into a fresh model / optimizer, file holds each buffer once."""
    from vitm_amd import training
    from vitm_amd.utils.checkpoint import load_checkpoint, save_checkpoint
    cfg = W.CONFIGS["tiny2"]
    a = _model(cfg, 21)
    opt_a, sched_a = training.reference_optimizer_and_scheduler(a, 1000)
    tr_a = training.Trainer(a, opt_a, sched_a, max_norm=1.0)
    for step in range(2):
        tr_a.step(*_batch(cfg, 4, 300 + step))
    path = str(tmp_path / "ckpt.pt")
    save_checkpoint(a, path, opt_a, sched_a)
    nbytes = sum(p.numel() for p in a.parameters()) * 4
    assert os.path.getsize(path) < 3.5 * nbytes, "parameters + two moments, each stored once"
    b = _model(cfg, 1)
    opt_b, sched_b = training.reference_optimizer_and_scheduler(b, 1000)
    tr_b = training.Trainer(b, opt_b, sched_b, max_norm=1.0)
    load_checkpoint(b, path, opt_b, sched_b)
    assert tr_b.params.intact(), "load_state_dict copies into the flat views"
    x, y = _batch(cfg, 4, 302)
    la, lb = tr_a.step(x, y), tr_b.step(x, y)
    assert la.item() == lb.item()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb), k


@pytest.mark.parametrize("bf16", [False, True])
def test_graphed_train_step_matches_the_eager_step(bf16):
    """GraphedTrainStep (two eager steps, then one capture and replays; the AdamW scalars of the reference's warm-up /
    cosine schedule rewritten in device memory before each replay) against Trainer.step on an identical model: same
    losses, same gradient norms, same parameters and optimizer state after six steps, bit for bit."""
    from vitm_amd import training, GraphedTrainStep
    cfg = W.CONFIGS["tiny2"]
    cd = torch.bfloat16 if bf16 else None
    a, b = _model(cfg, 33), _model(cfg, 33)
    opt_a, sched_a = training.reference_optimizer_and_scheduler(a, 1000)
    opt_b, sched_b = training.reference_optimizer_and_scheduler(b, 1000)
    tr_a = training.Trainer(a, opt_a, sched_a, max_norm=1.0, compute_dtype=cd)
    tr_b = training.Trainer(b, opt_b, sched_b, max_norm=1.0, compute_dtype=cd)
    g = GraphedTrainStep(tr_b, warmup=2)
    for step in range(6):
        x, y = _batch(cfg, 4, 500 + step)
        la, lb = tr_a.step(x, y), g.step(x, y)
        assert la.item() == lb.item(), f"loss of step {step}"
        assert tr_a.last_grad_norm.item() == tr_b.last_grad_norm.item(), f"gradient norm of step {step}"
        assert opt_a.param_groups[0]["lr"] == opt_b.param_groups[0]["lr"]
    assert g.graph is not None and g.calls == 6
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb), k
        assert torch.equal(opt_a.state[pa]["exp_avg_sq"], opt_b.state[pb]["exp_avg_sq"]), k
        assert float(opt_a.state[pa]["step"]) == float(opt_b.state[pb]["step"]) == 6
    with pytest.raises(RuntimeError):
        g.step(*_batch(cfg, 2, 1))          # another batch shape than the captured one


def test_flat_adamw_state_loads_into_torch_adamw_with_one_step_per_parameter():
    """The optimizer state a flat-mode Trainer saves must load into torch.optim.AdamW (the reference's optimizer) with
    a step counter per parameter: one shared counter would advance once per parameter per iteration there."""
    import io
    from vitm_amd import training
    cfg = W.CONFIGS["tiny2"]
    a = _model(cfg, 5)
    opt = training.AdamW(a.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    tr = training.Trainer(a, opt, None, max_norm=1.0)
    for step in range(2):
        tr.step(*_batch(cfg, 4, 300 + step))
    buf = io.BytesIO()
    torch.save(opt.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, weights_only=True)
    steps = [st["step"] for st in sd["state"].values()]
    assert all(float(t) == 2 for t in steps) and len({id(t) for t in steps}) == len(steps)
    b = _model(cfg, 5)
    topt = torch.optim.AdamW(b.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    topt.load_state_dict(sd)
    for p in b.parameters():
        p.grad = torch.zeros_like(p)
    topt.step()
    assert all(float(st["step"]) == 3 for st in topt.state.values())
    # and back: a checkpoint whose entries alias one step tensor is de-aliased on load
    shared = torch.tensor(7.0)
    for st in sd["state"].values():
        st["step"] = shared
    opt.load_state_dict(sd)
    tr.step(*_batch(cfg, 4, 310))
    assert all(float(st["step"]) == 8 for st in opt.state.values())


def test_trainer_with_a_frozen_parameter():
    """Fine-tuning with part of the model frozen: the frozen tensors get no gradient and do not move."""
    from vitm_amd import training
    cfg = W.CONFIGS["tiny2"]
    m = _model(cfg, 6)
    m.vit_model.pos_embed.requires_grad_(False)
    m.vit_model.blocks[0].attn.qkv.weight.requires_grad_(False)
    frozen = {k: p.detach().clone() for k, p in m.named_parameters() if not p.requires_grad}
    opt = training.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    tr = training.Trainer(m, opt, None, max_norm=1.0)
    l0 = tr.step(*_batch(cfg, 4, 400))
    l1 = tr.step(*_batch(cfg, 4, 400))
    assert torch.isfinite(l0) and l1 < l0
    for k, p in m.named_parameters():
        if k in frozen:
            assert torch.equal(p.detach(), frozen[k]) and p.grad is None


def test_gradient_reducer_on_rccl_single_rank():
    """The RCCL leg of the data-parallel step on the one GPU at hand: a one-rank nccl group with the reducer forced on
    (bucketed asynchronous all-reduces with ReduceOp.AVG over slices of the flat gradient buffer, the parameter broadcast,
    the waits before clipping).  An average over one replica is the identity, so three steps must reproduce the steps
    of a Trainer without a group bit for bit; a mis-ordered collective (reducing a slice before the backward kernels
    that fill it have run, or clipping before the reduction has landed) would not."""
    import socket
    import torch.distributed as dist
    from vitm_amd import training
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        cfg = W.CONFIGS["tiny2"]
        runs = []
        for single in (False, True):
            m = _model(cfg, 8)
            opt = training.AdamW(m.parameters(), lr=1e-3)
            tr = training.Trainer(m, opt, None, max_norm=1.0, group=dist.group.WORLD, bucket_bytes=64 << 10,
                                  reduce_single_rank=single)
            assert tr.reducer.active == single and (not single or tr.reducer.avg)
            losses = [tr.step(*_batch(cfg, 4, 500 + i)).item() for i in range(3)]
            runs.append((losses, [p.detach().clone() for p in m.parameters()]))
        assert runs[0][0] == runs[1][0]
        for a, b in zip(runs[0][1], runs[1][1]):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()
