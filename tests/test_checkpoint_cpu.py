"""Checkpoint compatibility with the reference's on-disk formats (host logic only: no GPU, no kernels).

The fixtures were written by the imported reference modules (tests/golden/make_golden_checkpoint.py): the
ModelCheckpoint format (callbacks.py:28-33) and the end-of-training format (train.py:245-249) with the AdamW /
SequentialLR states of getOptimizerAndSchedular (train.py:65-93) after two steps.
"""
import os

import torch
import torch.nn as nn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = dict(image_size=16, patch_size=8, num_classes=5, num_heads=2, mlp_ratio=2.0, depth=2, embed_dim=64)


def _model():
    from vitm_amd.models import VisionTransformer, VITC
    LN = nn.LayerNorm
    c = CFG
    vt = VisionTransformer(c["image_size"], patch_size=c["patch_size"], num_classes=c["num_classes"],
                           num_heads=c["num_heads"], mlp_ratio=c["mlp_ratio"], depth=c["depth"],
                           embed_dim=c["embed_dim"], norm_layer=LN, embed_norm_layer=LN, final_norm_layer=LN)
    return VITC(vt)


def _optim(model):
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.98))
    sched = torch.optim.lr_scheduler.SequentialLR(
        opt, [torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=500),
              torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=500, eta_min=1e-6)], milestones=[500])
    return opt, sched


def test_bare_state_dict_loads_strict_with_reference_key_order():
    from vitm_amd.utils.checkpoint import load_checkpoint, read_checkpoint
    m = _model()
    ck = load_checkpoint(m, os.path.join(GOLD, "ref_ckpt_best.pt"))
    ref = ck["model_state_dict"]
    mine = m.state_dict()
    assert list(mine.keys()) == list(ref.keys()), "parameter names and registration order must be the reference's"
    for k in ref:
        assert mine[k].shape == ref[k].shape and torch.equal(mine[k], ref[k]), k
    assert set(read_checkpoint(os.path.join(GOLD, "ref_ckpt_best.pt"))) == {"model_state_dict"}


def test_training_checkpoint_restores_model_optimizer_and_scheduler():
    from vitm_amd.utils.checkpoint import load_checkpoint
    m = _model()
    opt, sched = _optim(m)
    ck = load_checkpoint(m, os.path.join(GOLD, "ref_ckpt_final.pt"), optimizer=opt, scheduler=sched)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict"}
    # the reference's per-parameter Adam state lands on parameters of the same shape, in order
    ref_state = ck["optimizer_state_dict"]["state"]
    params = list(m.parameters())
    assert len(ref_state) == len(params)
    for i, p in enumerate(params):
        assert opt.state[p]["exp_avg"].shape == p.shape
        assert torch.equal(opt.state[p]["exp_avg"], ref_state[i]["exp_avg"])
        assert float(opt.state[p]["step"]) == 2.0
    g = opt.param_groups[0]
    assert g["weight_decay"] == 0.05 and tuple(g["betas"]) == (0.9, 0.98)
    assert sched.last_epoch == 2
    assert abs(g["lr"] - 1e-3 * (0.01 + 2 * 0.99 / 500)) < 1e-9     # LinearLR warm-up, step 2 (train.py:78-83)


def test_round_trip_both_formats(tmp_path):
    from vitm_amd.utils.checkpoint import load_checkpoint, read_checkpoint, save_checkpoint, save_state_dict
    m = _model()
    opt, sched = _optim(m)
    load_checkpoint(m, os.path.join(GOLD, "ref_ckpt_final.pt"), optimizer=opt, scheduler=sched)
    a, b = str(tmp_path / "best.pt"), str(tmp_path / "final.pt")
    save_state_dict(m, a)
    save_checkpoint(m, b, opt, sched)
    ref = torch.load(os.path.join(GOLD, "ref_ckpt_final.pt"), weights_only=True)
    mine = torch.load(b, weights_only=True)
    assert list(mine.keys()) == list(ref.keys())
    assert list(mine["model_state_dict"].keys()) == list(ref["model_state_dict"].keys())
    for k, v in ref["model_state_dict"].items():
        assert torch.equal(mine["model_state_dict"][k], v)
    assert mine["optimizer_state_dict"]["param_groups"][0]["params"] == ref["optimizer_state_dict"]["param_groups"][0]["params"]
    assert mine["scheduler_state_dict"]["last_epoch"] == ref["scheduler_state_dict"]["last_epoch"]
    m2 = _model()
    load_checkpoint(m2, a)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert set(read_checkpoint(a)) == {"model_state_dict"}


def test_rejects_non_checkpoint(tmp_path):
    import pytest
    from vitm_amd.utils.checkpoint import read_checkpoint
    p = str(tmp_path / "x.pt")
    torch.save({"a": 1}, p)
    with pytest.raises(ValueError):
        read_checkpoint(p)
