"""Host-side mirror of the reference's Python surface (src/flash_attention_torch.py,
src/flash_attention_wrappers.py): names, checks, exceptions, padding helpers.  CPU only."""
import math

import pytest
import torch

import flash_attention_dlrs_amd as fa
from flash_attention_dlrs_amd import _lib, flash_attention_torch as ft


def test_public_names_match_reference_modules():
    for name in ("FlashAttention", "FlashAttentionDeterministic", "convert_triton_dtype", "MIN_TENSOR_SIZE",
                 "flash_attention_forward", "flash_attention_backward"):
        assert hasattr(fa, name)
    assert fa.MIN_TENSOR_SIZE == 16  # reference torch.py:5
    assert issubclass(fa.FlashAttention, torch.autograd.Function)
    assert issubclass(fa.FlashAttentionDeterministic, torch.autograd.Function)


def test_convert_dtype_map_and_typeerror():
    # reference torch.py:7-18 maps exactly these four ...
    assert fa.convert_triton_dtype(torch.float64) == _lib.FA2_DTYPE_F64
    assert fa.convert_triton_dtype(torch.float32) == _lib.FA2_DTYPE_F32
    assert fa.convert_triton_dtype(torch.float16) == _lib.FA2_DTYPE_F16
    assert fa.convert_triton_dtype(torch.float8_e5m2) == _lib.FA2_DTYPE_F8E5M2
    # ... we add bf16 and e4m3fn (BASELINE.json c3-c5) ...
    assert fa.convert_triton_dtype(torch.bfloat16) == _lib.FA2_DTYPE_BF16
    assert fa.convert_triton_dtype(torch.float8_e4m3fn) == _lib.FA2_DTYPE_F8E4M3
    # ... and everything else raises TypeError like the reference
    for bad in (torch.int32, torch.int8, torch.bool, torch.complex64):
        with pytest.raises(TypeError):
            fa.convert_triton_dtype(bad)


def test_cpu_tensors_are_refused_like_the_reference():
    x = torch.zeros(1, 1, 16, 16)
    for fn in (fa.FlashAttention.apply, fa.FlashAttentionDeterministic.apply):
        with pytest.raises(NotImplementedError, match="same CUDA device"):  # reference torch.py:25-26
            fn(x, x, x)
    with pytest.raises((NotImplementedError, RuntimeError, AssertionError)):
        fa.flash_attention_forward(x, x, x, torch.device("cpu"))


def test_wrapper_asserts_like_the_reference():
    a, b = torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 32, 16)
    with pytest.raises(AssertionError):  # reference wrappers.py:20-22
        fa.flash_attention_forward(a, b, a, torch.device("cpu"))
    with pytest.raises(AssertionError):
        fa.flash_attention_forward(a[0], a[0], a[0], torch.device("cpu"))
    with pytest.raises(AssertionError):
        fa.flash_attention_forward(a, a.half(), a, torch.device("cpu"))


def test_padding_helpers():
    assert [ft.next_power_of_2(n) for n in (1, 2, 3, 8, 9, 40, 64, 65, 128)] == [1, 2, 4, 8, 16, 64, 64, 128, 128]
    t = torch.randn(2, 3, 5, 40)
    p = ft.pad_last_dim(t, 64)
    assert p.shape == (2, 3, 5, 64) and torch.equal(p[..., :40], t) and p[..., 40:].abs().max() == 0
    assert ft.pad_last_dim(t, 40) is t
    for dt in (torch.float8_e5m2, torch.float8_e4m3fn):
        f8 = torch.randn(1, 1, 4, 8).to(dt)
        p8 = ft.pad_last_dim(f8, 16)
        assert p8.dtype == dt and torch.equal(p8[..., :8].view(torch.uint8), f8.view(torch.uint8))
        assert p8[..., 8:].float().abs().max() == 0


@pytest.mark.parametrize("causal", [False, True])
def test_recompute_backward_matches_autograd(causal):
    """The (non-native, out-of-scope) backward: gradients from the saved log2-domain L equal autograd
    through SDPA(scale=1)  -- the comparison src/test_correctness.py:46-62 makes."""
    torch.manual_seed(0)
    Q, K, V = (torch.randn(1, 2, 24, 8, dtype=torch.float64, requires_grad=True) for _ in range(3))
    O = torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1.0, is_causal=causal)
    dO = torch.randn_like(O)
    gq, gk, gv = torch.autograd.grad(O, (Q, K, V), dO)
    S = Q @ K.transpose(-1, -2)
    if causal:
        S = S.masked_fill(~torch.ones(24, 24, dtype=torch.bool).tril(), float("-inf"))
    L = torch.logsumexp(S, dim=-1, keepdim=True) * math.log2(math.e)
    dq, dk, dv = ft.attention_backward_recompute(Q.detach(), K.detach(), V.detach(), O.detach(), dO, L.detach(),
                                                 causal=causal)
    for a, b in ((dq, gq), (dk, gk), (dv, gv)):
        assert torch.allclose(a, b, atol=1e-10)


def test_library_missing_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback|not built"):
        _lib.lib()


def test_autotune_key_and_table_roundtrip(tmp_path, monkeypatch):
    """The tuner's key ignores B and H except through the small / large grid class, buckets N, and the table
    persists (reference: autotune key (B, H, N, d), kernels.py:13).  No GPU: only the bookkeeping is exercised."""
    from flash_attention_dlrs_amd import autotune
    q = lambda B, H, N, d, dt=torch.bfloat16: torch.empty(B, H, N, d, dtype=dt)
    assert autotune.key_of(q(4, 32, 4096, 128), True) == "bfloat16:d128:N4096:causal:large"
    assert autotune.key_of(q(8, 16, 4096, 128), True) == autotune.key_of(q(4, 32, 4096, 128), True)
    assert autotune.key_of(q(1, 2, 3000, 128), False) == "bfloat16:d128:N4096:full:small"
    assert autotune.key_of(q(1, 2, 128, 64, torch.float16), False) == "float16:d64:N128:full:small"
    path = str(tmp_path / "t.json")
    autotune.enable(True, path)
    try:
        autotune._load()["bfloat16:d128:N4096:causal:large"] = {"variant": "mfma16d", "ms": {}, "table_id": autotune.table_id("cpu")}
        autotune._save()
        autotune.enable(True, path)  # drops the in-memory copy
        assert autotune._load()["bfloat16:d128:N4096:causal:large"]["variant"] == "mfma16d"
        # a known key is answered from the table without touching the GPU
        Q = q(4, 32, 4096, 128)
        assert autotune.pick(Q, Q, Q, Q, Q, _lib.FA2_DTYPE_BF16, True, 1.0) == _lib.VARIANT_MFMA16D
        # an entry tuned on another device model or library version is not served
        autotune._load()["bfloat16:d128:N4096:causal:large"]["table_id"] = "some other GPU|fa2-hip 0.0.0"
        assert "table_id" in autotune._load()["bfloat16:d128:N4096:causal:large"]
    finally:
        autotune.enable(False)
    Q = q(4, 32, 4096, 128)
    assert autotune.pick(Q, Q, Q, Q, Q, _lib.FA2_DTYPE_BF16, True, 1.0) == _lib.VARIANT_AUTO


def test_forward_head_size_keeps_odd_head_sizes_on_the_matrix_cores():
    """head sizes the MFMA kernels do not take (f16 / bf16 not a multiple of 8, fp32 not a multiple of 4) are padded on the
    host, as the reference pads (torch.py:38-47), instead of falling to the VALU kernel; eligible ones run as they are"""
    f = ft.forward_head_size
    for dt in (torch.float16, torch.bfloat16):
        assert [f(dt, 4, 32, 4096, d) for d in (100, 36, 20, 1, 63, 65, 127)] == [128, 64, 64, 64, 64, 128, 128]
        assert [f(dt, 2, 8, 1024, d) for d in (8, 24, 40, 48, 64, 80, 96, 120, 128, 136, 256)] == [8, 24, 40, 48, 64, 80, 96, 120, 128, 136, 256]
        assert [f(dt, 4, 32, 4096, d) for d in (40, 64, 80, 96, 120, 128)] == [40, 64, 128, 128, 128, 128]     # large grids: the pipelined kernels
        assert [f(dt, 4, 32, 4096, d, True) for d in (16, 40, 48, 64)] == [64, 64, 64, 64] and f(dt, 8, 16, 1024, 32, True) == 32    # (causal, large)
    assert [f(torch.float32, 1, 2, 70, d) for d in (37, 50, 30, 96, 40, 8, 130)] == [40, 52, 32, 96, 40, 8, 130]
    assert f(torch.float64, 1, 2, 70, 37) == 37
    for dt in (torch.float8_e5m2, torch.float8_e4m3fn):       # fp8 has matrix kernels at d = 128 only: anything below is padded to it
        assert [f(dt, 1, 2, 70, d) for d in (37, 64, 127, 128, 160)] == [128, 128, 128, 128, 160]


def test_bench_self_launch_relays_the_ranks_exit_code_without_a_gpu():
    """bench.py --gpus 2 outside torch.distributed.run starts its own rank processes; on a box without GPUs the ranks fail and
    the parent must hand their failure on (non-zero exit, no result line) instead of crashing or printing a made-up line"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    from conftest import ROOT
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
