"""The CPU oracle against the fixtures captured from the imported reference.

Bar: the oracle is a different op decomposition (im2col GEMM, explicit LN/softmax/GELU) of the
same fp32 arithmetic, so it must agree with the reference's outputs to fp32 round-off:
1e-4 relative to the tensor's max magnitude is the north-star fp32 tolerance; we hold 2e-5.
"""
import os

import numpy as np
import pytest
import torch

import weights as W
from oracle import vit_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


FIXTURE_CONFIG = {"vit_l16_d4": "vit_l16"}   # fixtures of a cut-depth network: file stem -> configuration


def _load(name):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    cfg = W.CONFIGS[FIXTURE_CONFIG.get(name, name)]
    depth = int(g["depth"])
    sd = {k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, int(g["seed"]), depth).items()}
    x = torch.from_numpy(W.make_input(cfg, int(g["batch"]), int(g["seed"]) + 1))
    return g, cfg, sd, x


@pytest.mark.parametrize("name", ["tiny2", "cifar_recipe"])
def test_oracle_stages_match_reference(name):
    g, cfg, sd, x = _load(name)
    stages = []
    emb = O.vit_forward(x, sd, W.oracle_cfg(cfg), "vit_model.", torch.float32, stages)
    assert _rel(stages[0].numpy(), g["features"]) < TOL
    for i in range(int(g["depth"])):
        assert _rel(stages[i + 1].numpy(), g[f"block{i}"]) < TOL, f"block {i}"
    assert _rel(emb.numpy(), g["embeddings"]) < TOL
    logits = O.vitc_forward(x, sd, W.oracle_cfg(cfg))
    assert _rel(logits.numpy(), g["logits"]) < TOL


@pytest.mark.parametrize("name", ["vit_tiny_cifar", "vit_s16", "vit_b16", "vit_b16_384", "vit_l16_d4"])
def test_oracle_logits_match_reference(name):
    g, cfg, sd, x = _load(name)
    logits, emb = O.vitc_forward(x, sd, dict(W.oracle_cfg(cfg), depth=int(g["depth"])), return_embeddings=True)
    assert logits.shape == (int(g["batch"]), cfg["num_classes"])
    assert _rel(logits.numpy(), g["logits"]) < TOL
    if "embeddings_rows" in g:
        emb = emb[:, torch.from_numpy(g["embeddings_rows"])]
    assert _rel(emb.numpy(), g["embeddings"]) < TOL


def test_oracle_fp64_agrees_with_fp32():
    g, cfg, sd, x = _load("tiny2")
    l32 = O.vitc_forward(x, sd, W.oracle_cfg(cfg), torch.float32)
    l64 = O.vitc_forward(x, sd, W.oracle_cfg(cfg), torch.float64)
    assert _rel(l32.numpy(), l64.numpy()) < 1e-5


def test_oracle_asserts_on_wrong_image_size():
    # patch_embeddings.py:113-114, tested by the reference in tests/test_patch_embed.py:35-42
    g, cfg, sd, x = _load("tiny2")
    with pytest.raises(AssertionError):
        O.vit_forward(x[:, :, :30, :30], sd, W.oracle_cfg(cfg), "vit_model.")


def test_flops_formula_matches_survey_numbers():
    for name, gf in (("vit_s16", 9.197), ("vit_b16", 35.128), ("vit_b16_384", 110.969), ("vit_l16", 123.109)):
        f = O.flops_per_image(W.shapes(W.CONFIGS[name])) / 1e9
        assert abs(f - gf) / gf < 2e-3, (name, f)


@pytest.mark.parametrize("name", ["attn_qknorm_hd64", "attn_qknorm_hd96", "attn_qkvbias_hd32"])
def test_oracle_attention_variants_match_reference(name):
    """Attention(qkv_bias / qk_norm) fixtures captured from the imported reference (models.py:32-35,54-55,82)."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    y = O.attention(torch.from_numpy(g["x"]), sd, int(g["heads"]), "")
    ref = torch.from_numpy(g["y"])
    assert (y - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("name", ["pblock_d128", "pblock_d192"])
def test_oracle_parallel_block_matches_reference(name):
    """ParallelBlock fixtures captured from the imported reference (parallel_block.py:193-221)."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    y = O.parallel_block(torch.from_numpy(g["x"]), sd, int(g["heads"]))
    ref = torch.from_numpy(g["y"])
    assert (y - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
