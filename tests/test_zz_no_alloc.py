"""vitm_cross_entropy_f32_noalloc (the entry point that allocates nothing) against vitm_cross_entropy_f32, through the
C ABI: loss and gradient bit for bit, two calls in a row, the second on a smaller batch (the built-in scratch keeps the
first call's rows), and the size limit."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def test_cross_entropy_noalloc_matches_the_allocating_entry_point():
    from vitm_amd import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(5)
    for B, C in ((300, 1000), (37, 10)):          # the second call is smaller: rows 37.. of the scratch are stale
        logits = (torch.randn(B, C, generator=g) * 3).cuda()
        labels = torch.randint(0, C, (B,), generator=g).cuda()
        ref_loss = torch.zeros(1, device="cuda")
        ref_d = torch.full((B, C), float("nan"), device="cuda")
        rows = torch.empty(B, device="cuda")
        _lib.check(lib.vitm_cross_entropy_f32(_p(logits), C, _p(labels), B, C, _p(ref_loss), _p(ref_d), C, _p(rows), st), "ce")
        loss = torch.zeros(1, device="cuda")
        d = torch.full((B, C), float("nan"), device="cuda")
        _lib.check(lib.vitm_cross_entropy_f32_noalloc(_p(logits), C, _p(labels), B, C, _p(loss), _p(d), C, st), "ce noalloc")
        torch.cuda.synchronize()
        assert torch.equal(loss, ref_loss) and torch.equal(d, ref_d)
        want = torch.nn.functional.cross_entropy(logits.double(), labels)
        assert abs(loss.item() - want.item()) <= 1e-5 * max(1.0, abs(want.item()))
    too_many = lib.vitm_cross_entropy_noalloc_max_rows() + 1
    rc = lib.vitm_cross_entropy_f32_noalloc(_p(logits), C, _p(labels), too_many, C, _p(loss), None, C, st)
    assert rc != 0 and b"exceeds" in lib.vitm_last_error()


def test_cross_entropy_rejects_labels_out_of_range():
    """A label outside [0, C) (ignore_index -100 included) is a NaN loss and a zero gradient row, not a wild read."""
    from vitm_amd import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, C = 6, 10
    logits = torch.randn(B, C, device="cuda")
    labels = torch.tensor([1, 2, -100, 3, 10, 4], device="cuda")
    loss = torch.zeros(1, device="cuda")
    d = torch.full((B, C), float("nan"), device="cuda")
    _lib.check(lib.vitm_cross_entropy_f32_noalloc(_p(logits), C, _p(labels), B, C, _p(loss), _p(d), C, st), "ce")
    torch.cuda.synchronize()
    assert torch.isnan(loss).all()
    assert (d[2] == 0).all() and (d[4] == 0).all() and torch.isfinite(d).all()
