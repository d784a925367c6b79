"""CPU-only checks (no GPU needed): the C-ABI library loads and exports everything the header declares, argument
validation answers without touching a device, the module API mirrors the reference's names / attributes /
state_dict layout, and the host-side helpers behave like the reference's."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn as nn

import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vitm_gfx950.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vitm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vitm_amd import _lib
    names = _declared_functions()
    assert len(names) >= 12
    lib = _lib.load()
    for n in names:
        assert n in _lib.SIGNATURES, f"{n} declared in the header but not bound"
        assert hasattr(lib, n), f"{n} not exported by {_lib.LIB_PATH}"
    assert sorted(_lib.SIGNATURES) == names, "binding table and header disagree"


def test_version_and_constants_without_a_gpu():
    from vitm_amd import _lib, library_version
    assert "gfx950" in library_version()
    assert _lib.load().vitm_gemm_k_multiple() == 64


def test_argument_errors_are_reported_before_any_launch():
    from vitm_amd import _lib
    lib = _lib.load()
    rc = lib.vitm_gemm(None, 64, None, 64, None, 64, 0, 8, 8, 64, 0, None, None, None, None, None, 0, 0, None)
    assert rc == -1 and b"null" in lib.vitm_last_error()
    buf = ctypes.create_string_buffer(4096)  # a host buffer: only its (aligned) address is inspected
    addr = (ctypes.addressof(buf) + 255) & ~255
    p = ctypes.c_void_p(addr)
    rc = lib.vitm_gemm(p, 100, p, 100, p, 8, 0, 8, 8, 100, 0, None, None, p, None, None, 0, 0, None)
    assert rc == -1 and b"multiple of 64" in lib.vitm_last_error()
    rc = lib.vitm_attention(p, 3 * 136, p, 136, 1, 4, 1, 136, ctypes.c_float(0.1), None)
    assert rc == -3 and b"head_dim" in lib.vitm_last_error()
    rc = lib.vitm_ln_stats(p, 0, 16384, 4, 16384, ctypes.c_float(1e-5), p, None)
    assert rc == -1 and b"8192" in lib.vitm_last_error()
    rc = lib.vitm_patch_embed(p, 1, 1, 3, 32, 32, 64, 64, p, 64, 64, p, p, 8, 8, 1, None)
    assert rc == -1


def _build(cfg):
    from vitm_amd.models import VisionTransformer, VITC
    LN = nn.LayerNorm
    vt = VisionTransformer(cfg["image_size"], patch_size=cfg["patch_size"], num_classes=cfg["num_classes"],
                           num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"], depth=cfg["depth"],
                           embed_dim=cfg["embed_dim"], norm_layer=LN,
                           embed_norm_layer=LN if cfg["embed_norm"] else None,
                           final_norm_layer=LN if cfg["final_norm"] else None)
    return VITC(vt)


@pytest.mark.parametrize("name", ["tiny2", "vit_tiny_cifar", "cifar_recipe", "vit_s16"])
def test_state_dict_layout_is_the_reference_s(name):
    """weights.make_state_dict's keys/shapes were strict-loaded into the imported reference by make_golden.py,
    so matching them matches the reference's VITC.state_dict()."""
    cfg = W.CONFIGS[name]
    m = _build(cfg)
    ref = W.make_state_dict(cfg, 0)
    sd = m.state_dict()
    assert set(sd) == set(ref)
    for k, v in ref.items():
        assert tuple(sd[k].shape) == v.shape, k
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_public_names_and_attributes():
    import vitm_amd
    from vitm_amd.models import VisionTransformer, VITC, Attention, PatchEmbed, MLP, Mlp, Block, LayerScale
    from vitm_amd.models.models import Block as B2, MLP as M2, LayerScale as L2
    assert (B2, M2, L2) == (Block, MLP, LayerScale) and Mlp is MLP
    assert vitm_amd.models.__all__[:4] == ["VisionTransformer", "VITC", "Attention", "PatchEmbed"]
    pe = PatchEmbed(image_size=224, patch_size=16)
    assert pe.patch_size == (16, 16) and pe.image_size == (224, 224) and pe.grid_size == [14, 14]
    assert pe.num_patches == 196 and pe.flatten and pe.output_fmt is None
    assert pe.projection.bias is None and isinstance(pe.norm_layer, nn.Identity)
    a = Attention(768, num_heads=12)
    assert a.head_dim == 64 and a.scale == 64 ** -0.5 and a.qkv.bias is None and a.proj.bias is not None
    assert isinstance(a.q_norm, nn.Identity) and not a.fused_attention and not a.qk_norm
    mlp = MLP(192, hidden_features=100, out_features=7, bias=True)
    assert mlp.out_features == 192 and mlp.bias == (True, True)     # out_features is ignored, as in the reference
    assert MLP(64).hidden_features == 256 and MLP(64).linear_layer1.bias is None
    blk = Block(192, 3, 4.0)
    assert blk.mlp.hidden_features == 768 and blk.attn.qkv.bias is None and blk.mlp.linear_layer2.bias is not None
    assert torch.allclose(blk.ls1.gamma, torch.full((192,), 1e-5))
    vt = VisionTransformer(32, 4, 10, 3, 4.0, depth=2, embed_dim=192, norm_layer=nn.LayerNorm,
                           final_norm_layer=nn.LayerNorm)
    for attr in ("embed_dim", "num_classes", "image_size", "patch_size", "patch_embed", "class_token", "pos_embed",
                 "norm_pre", "blocks", "norm_final", "feature_info", "parallel_block", "shard_along_dims"):
        assert hasattr(vt, attr), attr
    assert vt.pos_embed.shape == (1, 65, 192) and vt.class_token.shape == (1, 1, 192)
    assert vt.feature_info == [dict(module="block_0", num_chs=192), dict(module="block_1", num_chs=192)]
    assert VITC(vt).head.out_features == 10


def test_reference_quirks_are_kept_or_fixed_as_documented():
    from vitm_amd.models import VisionTransformer
    vt = VisionTransformer(32, 4, 10, 3, 4.0, depth=1, embed_dim=192, norm_layer=nn.LayerNorm, class_token=False)
    assert vt.class_token is not None                      # models.py:299 tests `is not None` on a bool
    vt = VisionTransformer(32, 4, 10, 3, 4.0, depth=1, embed_dim=192)  # norm_layer=None: superset behaviour
    assert isinstance(vt.norm_pre, nn.Identity) and isinstance(vt.blocks[0].norm1, nn.LayerNorm)
    assert vt.patch_embed.projection.out_channels == 192   # embed_dim reaches PatchEmbed
    vt = VisionTransformer(32, 4, 10, 3, 0.7, depth=1, embed_dim=192, parallel_block=True)   # fixed constructor call
    assert type(vt.blocks[0]).__name__ == "ParallelBlock" and vt.blocks[0].mlp_hidden_dim == int(192 * 0.7)


def test_initialisation_law():
    from vitm_amd.models import VisionTransformer
    torch.manual_seed(0)
    vt = VisionTransformer(32, 4, 10, 3, 4.0, depth=2, embed_dim=192, norm_layer=nn.LayerNorm)
    for m in vt.modules():
        if isinstance(m, nn.Linear):
            bound = (6.0 / (m.in_features + m.out_features)) ** 0.5
            assert m.weight.abs().max() <= bound + 1e-6 and m.weight.std() > 0.4 * bound
            assert m.bias is None or torch.count_nonzero(m.bias) == 0
        elif isinstance(m, nn.LayerNorm):
            assert torch.all(m.weight == 1) and torch.all(m.bias == 0)
    assert 0.015 < vt.pos_embed.std() < 0.025 and 0.005 < vt.class_token.std() < 0.04   # trunc_normal(std=0.02)


def test_helpers_match_reference_semantics():
    from vitm_amd.utils.helper import to_2tuple, Format, nchw_to
    assert to_2tuple(4) == (4, 4) and to_2tuple((3, 5)) == (3, 5) and to_2tuple([7, 8]) == (7, 8)
    assert to_2tuple(True) == (True, True) and to_2tuple("ab") == ("ab", "ab")
    x = torch.arange(2 * 3 * 4 * 5.0).reshape(2, 3, 4, 5)
    assert nchw_to(x, Format.NHWC).shape == (2, 4, 5, 3) and nchw_to(x, "NHWC").shape == (2, 4, 5, 3)
    assert nchw_to(x, Format.NLC).shape == (2, 20, 3) and nchw_to(x, Format.NCL).shape == (2, 3, 20)
    assert nchw_to(x, Format.NCHW) is x and nchw_to(x, None) is x
    assert torch.equal(nchw_to(x, Format.NLC), x.flatten(2).transpose(1, 2))


def test_cpu_tensors_are_refused_not_computed_elsewhere():
    from vitm_amd.models import PatchEmbed, Attention, MLP, Block, LayerScale
    m = _build(W.CONFIGS["tiny2"])
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.vit_model.forward_features(torch.zeros(1, 3, 32, 32))
    for mod, x in ((PatchEmbed(32, 8, embed_dim=64), torch.zeros(1, 3, 32, 32)), (Attention(64, 2), torch.zeros(1, 5, 64)),
                   (MLP(64), torch.zeros(1, 5, 64)), (Block(64, 2, 2.0), torch.zeros(1, 5, 64)),
                   (LayerScale(64), torch.zeros(1, 5, 64))):
        with pytest.raises(RuntimeError, match="no CPU path"):
            mod(x)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from vitm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.VitmLibraryError, match="no fallback"):
        _lib.load()


def test_shard_bounds_partition_the_batch():
    from vitm_amd.sharding import shard_bounds, all_shard_bounds
    for n in (0, 1, 7, 8, 9, 256, 1001):
        for w in (1, 2, 3, 4, 8):
            b = all_shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def test_parallel_block_state_dict_matches_reference_fixture():
    """Names, shapes and order of ParallelBlock's parameters against a state_dict dumped from the reference module."""
    import os
    import numpy as np
    from vitm_amd.models import ParallelBlock, VisionTransformer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pblock_d192.npz"))
    ref = [(k[3:], g[k].shape) for k in g.files if k.startswith("sd.")]
    m = ParallelBlock(192, 3, mlp_ratio=float(g["mlp_ratio"]), mlp_bias=bool(g["mlp_bias"]))
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in ref]
    m.load_state_dict({k[3:]: __import__("torch").from_numpy(g[k]) for k in g.files if k.startswith("sd.")}, strict=True)
    vt = VisionTransformer(32, patch_size=8, num_classes=3, num_heads=2, mlp_ratio=0.7, depth=2, embed_dim=128,
                           parallel_block=True)
    assert all(isinstance(b, ParallelBlock) for b in vt.blocks) and vt.parallel_block
    assert "blocks.1.fused_proj_2.weight" in vt.state_dict()


def test_bench_reports_counter_files_only_for_the_sources_they_were_collected_on(tmp_path, monkeypatch):
    """bench.py's PMC figures come from committed files; a file collected on other kernel sources (another csrc digest)
    or for another workload must be withheld, with the reason in the note."""
    import importlib
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    (tmp_path / "vision-transformer_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "profiles").mkdir()
    (tmp_path / "vision-transformer_amd" / "csrc" / "k.hip").write_text("// kernel v1\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    digest = bench.csrc_digest()
    rec = {"config": "vit_b16", "batch_per_gpu": 256, "commit": "abc1234", "csrc_sha256": digest, "hbm_bytes_per_launch": 123.0}
    (tmp_path / "profiles" / "r3_fc1_traffic.json").write_text(json.dumps(rec))
    val, note = bench.committed_counters("fc1", "vit_b16", 256)
    assert val == 123.0 and "abc1234" in note
    val, note = bench.committed_counters("fc1", "vit_s16", 256)          # other workload
    assert val is None and "no committed" in note
    (tmp_path / "vision-transformer_amd" / "csrc" / "k.hip").write_text("// kernel v2\n")   # sources moved on
    val, note = bench.committed_counters("fc1", "vit_b16", 256)
    assert val is None and "stale" in note
    # the committed files of this repository carry the provenance fields
    import glob
    for path in glob.glob(os.path.join(root, "profiles", "r3_*traffic.json")):
        t = json.load(open(path))
        assert t.get("csrc_sha256") and "commit" in t, path
