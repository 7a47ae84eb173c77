"""GPU parity tests: the HIP path, called through the C ABI (libfa2_hip.so via the Python surface),
against (a) the committed golden vectors produced by the reference kernel itself, (b) the CPU oracle
on seeded inputs at small sizes, (c) live torch SDPA(scale=1) on the same device at BASELINE.json's
full sizes -- the comparison the reference's own test makes (src/test_correctness.py:33-40) -- and
size-independent properties there.

Tolerances (stated once):
  fp32      allclose(atol=1e-4, rtol=1e-5)  -- the reference's own bar (test_correctness.py:40);
            max-abs <= 1e-3 (north_star)
  fp16      |O - ref| <= 6e-3   (P and O carry 11 significant bits; 1.5 ulp of an |O| in [4, 8))
  bf16      |O - ref| <= 5e-2   (8 significant bits; 1.5 ulp of an |O| in [4, 8): the oracle and the kernel each
            round P relative to their own running max and round O once more)
  fp8       >= 95 % of elements equal to the oracle's fp8 value, the rest within one fp8 ulp
  L         fp32 5e-5 * max(1,|L|) ; fp16/bf16 one ulp of the dtype at |L|
"""
import json
import math

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

import flash_attention_dlrs_amd as fa  # noqa: E402
from flash_attention_dlrs_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
O_TOL = {torch.float32: 1e-4, torch.float16: 6e-3, torch.bfloat16: 5e-2, torch.float64: 1e-9}
ORACLE_NAME = {torch.float32: "float32", torch.float16: "float16", torch.bfloat16: "bfloat16",
               torch.float64: "float64", torch.float8_e5m2: "float8_e5m2", torch.float8_e4m3fn: "float8_e4m3fn"}


def hip_forward(Q, K, V, causal=False, scale=1.0, variant="auto"):
    O, L = fa.flash_attention_forward(Q.to(DEV), K.to(DEV), V.to(DEV), DEV, causal=causal, scale=scale,
                                      variant=variant)
    torch.cuda.synchronize()
    return O.cpu(), L.cpu()


def supported_variants(dtype, d):
    v = ["auto", "generic"]
    if dtype in (torch.float16, torch.bfloat16) and d in (64, 128):
        # every variant include/fa2_fwd.h publishes (a64 has its own file: it needs N % 256 == 0, tests/test_a64_parity.py)
        v += ["mfma16", "mfma16_w8", "mfma16d", "mfma16d_w4", "mfma16h", "mfma16h_w4", "mfma16k", "mfma16k_r2k2"]
        if d == 128:
            pass
        else:
            v += ["mfma16k_r2k4"]
    if dtype == torch.float32 and d in (64, 128):
        v += ["mfma32"]
    return v


def ulp(dtype, x):
    mant = {torch.float16: 10, torch.bfloat16: 7}[dtype]
    return 2.0 ** (math.floor(math.log2(max(abs(x), 1e-30))) - mant)


def check_O(O, O_ref, dtype):
    if dtype == torch.float32:
        assert torch.allclose(O_ref, O, atol=1e-4, rtol=1e-5)
    assert (O.float() - O_ref).abs().max() <= O_TOL[dtype]


def check_L(L, L_ref, dtype):
    L, L_ref = L.double().flatten(), torch.as_tensor(L_ref).double().flatten()
    if dtype in (torch.float32, torch.float64):
        assert ((L - L_ref).abs() <= 5e-5 * L_ref.abs().clamp(min=1)).all()
    else:
        assert (L - L_ref).abs().max() <= 1.01 * ulp(dtype, L_ref.abs().max().item())


# ----------------------------------------------------------------------------- (a) golden vectors
@pytest.mark.parametrize("name,tile", [("c1_f32_seed0", "32x64"), ("c1_f32_seed1", "64x32")])
def test_golden_c1_fp32(name, tile):
    g = load_golden(name)
    Q, K, V = (torch.from_numpy(g[k]) for k in "QKV")
    for variant in supported_variants(torch.float32, 64):
        O, L = hip_forward(Q, K, V, variant=variant)
        O_ref, O_sdpa = torch.from_numpy(g[f"O_ref_{tile}"]), torch.from_numpy(g["O_sdpa"])
        assert torch.allclose(O_sdpa, O, atol=1e-4, rtol=1e-5), variant   # reference's own bar
        assert (O - O_sdpa).abs().max() <= 1e-3
        assert (O - O_ref).abs().max() < 3e-5, variant                     # vs the reference kernel's output
        check_L(L, g[f"L_ref_{tile}"], torch.float32)


def test_golden_fp16():
    g = load_golden("c1_f16_seed2")
    Q, K, V = (torch.from_numpy(g[k]) for k in "QKV")
    for variant in supported_variants(torch.float16, 64):
        O, L = hip_forward(Q, K, V, variant=variant)
        assert O.dtype == torch.float16 and L.dtype == torch.float16
        assert (O.float() - torch.from_numpy(g["O_ref_32x32"]).float()).abs().max() <= 2 ** -8, variant  # 2 ulp at |O|<4
        assert (O.float() - torch.from_numpy(g["O_sdpa"])).abs().max() <= O_TOL[torch.float16]
        check_L(L, g["L_ref_32x32"].astype(np.float32), torch.float16)


def test_golden_d128_default_kernel_against_the_reference_kernel_itself():
    """the default 16-bit kernel (a64: d = 128, N >= 256) against a vector the REFERENCE kernel produced
    (fp16, (1,1,512,128), tile 64x64 under the interpreter) -- directly, not through the oracle: <= 2 fp16 ulp of O, L one ulp"""
    g = load_golden("d128_f16_n512_seed15")
    Q, K, V = (torch.from_numpy(g[k]) for k in "QKV")
    O_ref = torch.from_numpy(g["O_ref_64x64"]).float()
    for variant in ("a64", "auto", "mfma16h_w4", "mfma16d_w4", "mfma16k", "mfma16", "generic"):
        O, L = hip_forward(Q, K, V, variant=variant)
        assert O.dtype == torch.float16 and L.dtype == torch.float16
        # 2 fp16 ulp at the outputs' scale (|O| in [1, 2): 2^-9).  The kernels round P relative to a deferred running maximum
        # and sum in another order than the reference's 64x64 tiles, so bits differ where a sum sits near a rounding boundary:
        # the oracle's deferred mode (CPU suite) is bit-equal to this vector on 92 % of the elements, max |diff| 2^-9
        err = (O.float() - O_ref).abs()
        assert err.max() <= 2.0 ** -9, (variant, err.max().item())
        assert (err <= 2.0 * torch.tensor([ulp(torch.float16, x) for x in O_ref.flatten().tolist()]).reshape(O_ref.shape).clamp(min=2.0 ** -11)
                ).float().mean() > 0.999, variant
        assert (O.float() == O_ref).float().mean() > 0.85, variant
        assert (O.float() - torch.from_numpy(g["O_sdpa"])).abs().max() <= O_TOL[torch.float16], variant
        check_L(L, g["L_ref_64x64"].astype(np.float32), torch.float16)


def test_golden_d128_fp32_reference_test_head_size():
    """fp32 at the head size of the reference's own test (src/test_correctness.py:9-14), against the reference kernel's output"""
    g = load_golden("d128_f32_n256_seed16")
    Q, K, V = (torch.from_numpy(g[k]) for k in "QKV")
    for variant in supported_variants(torch.float32, 128):
        O, L = hip_forward(Q, K, V, variant=variant)
        assert torch.allclose(torch.from_numpy(g["O_sdpa"]), O, atol=1e-4, rtol=1e-5), variant
        assert (O - torch.from_numpy(g["O_ref_32x32"])).abs().max() < 3e-5, variant
        check_L(L, g["L_ref_32x32"], torch.float32)


def test_golden_bf16_and_causal():
    g = load_golden("c1_bf16_seed4")
    Q, K, V = (torch.from_numpy(g[k].view(np.int16)).view(torch.bfloat16) for k in "QKV")
    for variant in supported_variants(torch.bfloat16, 64):
        for causal, key in ((False, "O_sdpa"), (True, "O_sdpa_causal")):
            O, _ = hip_forward(Q, K, V, causal=causal, variant=variant)
            assert (O.float() - torch.from_numpy(g[key])).abs().max() <= O_TOL[torch.bfloat16], (variant, causal)
    g = load_golden("c1_f32_causal_seed3")
    Q, K, V = (torch.from_numpy(g[k]) for k in "QKV")
    for variant in supported_variants(torch.float32, 64):
        O, _ = hip_forward(Q, K, V, causal=True, variant=variant)
        assert torch.allclose(torch.from_numpy(g["O_sdpa"]), O, atol=1e-4, rtol=1e-5), variant


@pytest.mark.parametrize("name,d", [("pad_d40_f32_seed5", 40), ("pad_d8_f32_seed6", 8)])
def test_golden_padding(name, d, monkeypatch):
    """the reference's padding cases (torch.py:38-47) WITHOUT the three host pad copies: the kernels
    take d = 40 / d = 8 as they are -- the forward must not call the padding helper at all"""
    from flash_attention_dlrs_amd import flash_attention_torch as ft, flash_attention_wrappers as fw

    def no_pad(*a, **k):
        raise AssertionError("the forward path padded a tensor on the host")
    monkeypatch.setattr(ft, "pad_last_dim", no_pad)
    monkeypatch.setattr(fw, "pad_last_dim", no_pad)
    g = load_golden(name)
    Q, K, V = (torch.from_numpy(g[k]) for k in "QKV")
    assert _lib.query_tile(Q.shape[2], d, _lib.FA2_DTYPE_F32, False, B=Q.shape[0], H=Q.shape[1])[0] == _lib.VARIANT_MFMA32
    O_dev, _ = fa.flash_attention_forward(Q.to(DEV), K.to(DEV), V.to(DEV), DEV)
    assert O_dev.shape[-1] == d and O_dev.is_contiguous()   # exactly d columns (the reference returns a view of a padded O)
    O, L = hip_forward(Q, K, V)
    assert (O - torch.from_numpy(g["O_ref"])).abs().max() < 3e-5
    assert torch.allclose(torch.from_numpy(g["O_sdpa"]), O, atol=1e-4, rtol=1e-5)
    check_L(L, g["L_ref"], torch.float32)
    O2 = fa.FlashAttention.apply(Q.to(DEV), K.to(DEV), V.to(DEV)).cpu()  # the autograd surface launches the same way
    assert torch.equal(O2, O)
    Og, Lg = hip_forward(Q, K, V, variant="generic")   # and the catch-all kernel loops to d
    assert (Og - torch.from_numpy(g["O_ref"])).abs().max() < 3e-5
    check_L(Lg, g["L_ref"], torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [8, 24, 40, 48, 80, 96, 120])
@pytest.mark.parametrize("causal", [False, True])
def test_head_sizes_that_are_not_powers_of_two_vs_oracle(oracle, dtype, d, causal):
    """d predication in the MFMA kernels (16-byte chunks past d zero-filled on load, never stored): canaries behind every
    row of O catch a store past d; ragged N on top.  Also the 8-wave form and the generic kernel."""
    B, H, N = 2, 3, 200
    Q, K, V = _rand((B, H, N, d), dtype, seed=100 + d, spread=0.7)
    O_ref, L_ref = _oracle(oracle, Q, K, V, dtype, causal)
    variants = ["auto", "generic"] + (["mfma32"] if dtype == torch.float32 else ["mfma16", "mfma16_w8"])
    assert _lib.query_tile(N, d, fa.convert_triton_dtype(dtype), causal, B=B, H=H)[0] in (_lib.VARIANT_MFMA32, _lib.VARIANT_MFMA16)
    for v in variants:
        # O rows are d wide inside a wider canary-filled arena: a column past d must stay untouched
        arena = torch.full((B, H, N, d + 8), 768.0, dtype=dtype, device=DEV)   # (exact in bf16)
        Ov = arena[..., :d]
        Lv = torch.empty(B, H, N, 1, dtype=dtype, device=DEV)
        _lib.fa2_fwd(Q.to(DEV), K.to(DEV), V.to(DEV), Ov, Lv, fa.convert_triton_dtype(dtype), causal=causal, variant=_lib.VARIANTS[v])
        torch.cuda.synchronize()
        assert (arena[..., d:].float() == 768.0).all(), v
        check_O(Ov.cpu(), O_ref, dtype)
        check_L(Lv.cpu(), L_ref, dtype)


def test_odd_head_sizes(oracle):
    """head sizes no MFMA kernel takes: at the C ABI they run on the generic kernel; the Python surface pads them on the host (as
    the reference does, torch.py:38-47) so that they stay on the matrix cores, and returns the [:d] view"""
    assert _lib.query_tile(70, 40, _lib.FA2_DTYPE_F8E5M2, False, B=1, H=2)[0] == _lib.VARIANT_GENERIC
    for dtype, d in ((torch.float32, 37), (torch.float16, 20), (torch.bfloat16, 1), (torch.float32, 130), (torch.bfloat16, 100)):
        Q, K, V = _rand((1, 2, 70, d), dtype, seed=d, spread=0.5)
        assert _lib.query_tile(70, d, fa.convert_triton_dtype(dtype), False, B=1, H=2)[0] == _lib.VARIANT_GENERIC
        O_ref, L_ref = _oracle(oracle, Q, K, V, dtype, False)
        for variant in ("auto", "generic"):      # (a forced variant gets the tensors as they are)
            O, L = hip_forward(Q, K, V, variant=variant)
            assert O.shape == Q.shape
            check_O(O, O_ref, dtype)
            check_L(L, L_ref, dtype)
        Oa = fa.FlashAttention.apply(Q.to(DEV), K.to(DEV), V.to(DEV)).cpu()
        check_O(Oa, O_ref, dtype)
    # fp8 below d = 128: padded to 128, where the fp8 matrix kernels are (the generic kernel on the tensors as they are gives the same
    # within the fp8 bars: the padded columns are zeros)
    for dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        for shape, causal in (((1, 2, 320, 64), False), ((2, 24, 1024, 80), True)):
            Q, K, V = _rand(shape, dtype, seed=shape[3], spread=0.5)
            O, L = hip_forward(Q, K, V, causal=causal)
            Og, Lg = hip_forward(Q, K, V, causal=causal, variant="generic")
            assert O.shape == Q.shape and O.dtype == dtype
            step = 0.25 if dtype == torch.float8_e5m2 else 0.125
            assert ((O.float() - Og.float()).abs() <= step * Og.float().abs() + 0.5 * step * V.float().abs().max()).all()
            assert ((L.float() - Lg.float()).abs() <= step * Lg.float().abs() + 1.5 * step).all()
    # large grid, 64 < d < 128 a multiple of 8: padded to 128 (the pipelined kernels); the result is the predicated kernel's
    Q, K, V = _rand((4, 32, 1024, 96), torch.bfloat16, seed=96, spread=0.5)
    O, _ = hip_forward(Q, K, V, causal=True)
    O2, _ = hip_forward(Q, K, V, causal=True, variant="mfma16")
    assert O.shape == Q.shape and (O.float() - O2.float()).abs().max() <= 2 * O_TOL[torch.bfloat16]


def test_golden_strided_inputs_and_stride_inheritance():
    g = load_golden("strided_bnhd_f32_seed7")
    Q, K, V = (torch.from_numpy(g[k + "_storage"]).to(DEV).transpose(1, 2) for k in "QKV")
    assert not Q.is_contiguous()
    O = fa.FlashAttention.apply(Q, K, V)
    assert O.stride() == Q.stride()  # O = empty_like(Q) inherits the permuted strides (reference torch.py:50)
    assert (O.cpu() - torch.from_numpy(g["O_ref"])).abs().max() < 3e-5
    O2, L2 = fa.flash_attention_forward(Q, K, V, DEV)
    assert O2.is_contiguous()         # wrapper path allocates contiguous O (reference wrappers.py:37)
    assert torch.equal(O2.cpu(), O.cpu())
    check_L(L2.cpu(), g["L_ref"], torch.float32)


@pytest.mark.parametrize("name", ["n16_f32_seed8", "n48_f32_seed9"])
def test_golden_small_N(name):
    g = load_golden(name)
    O, L = hip_forward(*(torch.from_numpy(g[k]) for k in "QKV"))
    assert (O - torch.from_numpy(g["O_ref"])).abs().max() < 3e-5
    check_L(L, g["L_ref"], torch.float32)


# ----------------------------------------------------------------------------- (b) seeded vs oracle
def _rand(shape, dtype, seed, spread=1.0):
    gen = torch.Generator().manual_seed(seed)
    return tuple((torch.randn(*shape, generator=gen) * spread).to(dtype) for _ in range(3))


def _oracle(oracle, Q, K, V, dtype, causal, scale=1.0):
    f = (lambda t: t.double().numpy()) if dtype == torch.float64 else (lambda t: t.float().numpy())
    t = _tile(Q.shape[2])
    O, L = oracle.forward(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, scale=scale, B_r=t, B_c=t)
    return torch.from_numpy(O), torch.from_numpy(L)


def _tile(N):
    for t in (64, 32, 16, 8, 4, 2, 1):
        if N % t == 0:
            return t


SHAPES = [(1, 1, 16, 16), (2, 3, 48, 32), (1, 2, 128, 64), (2, 2, 256, 128), (1, 2, 64, 256),
          (1, 2, 1, 64), (1, 1, 17, 64), (2, 1, 100, 128), (1, 2, 130, 64), (1, 1, 321, 128), (1, 2, 200, 32)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_seeded_vs_oracle(oracle, dtype, causal, shape):
    Q, K, V = _rand(shape, dtype, seed=sum(shape) + int(causal))
    O_ref, L_ref = _oracle(oracle, Q, K, V, dtype, causal)
    O64, _ = oracle.sdpa_f64(Q.float().numpy(), K.float().numpy(), V.float().numpy(), causal=causal)
    for variant in supported_variants(dtype, shape[-1]):
        O, L = hip_forward(Q, K, V, causal=causal, variant=variant)
        assert O.shape == Q.shape and L.shape == (*shape[:3], 1)
        tol = O_TOL[dtype]
        if dtype == torch.float32:
            assert torch.allclose(O_ref, O, atol=1e-4, rtol=1e-5), variant
        assert (O.float() - O_ref).abs().max() <= tol, variant
        assert (O.double() - torch.from_numpy(O64)).abs().max() <= (1e-3 if dtype == torch.float32 else tol), variant
        check_L(L, L_ref, dtype)


BIG_SHAPES = [(1, 3, 1000, 128), (2, 4, 777, 64), (1, 8, 1536, 128), (1, 16, 640, 128), (3, 8, 513, 64)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", BIG_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_ragged_multi_tile_shapes_vs_oracle(oracle, dtype, causal, shape):
    """Several Q tiles per (b, h) with ragged N: odd and even tile counts (causal tile pairs with and without a
    middle tile), B*H a multiple of 8 (XCD group mapping) and not, tails in both the Q and the K direction."""
    Q, K, V = _rand(shape, dtype, seed=sum(shape) * 3 + int(causal))
    O_ref, L_ref = _oracle(oracle, Q, K, V, dtype, causal)
    variants = ["auto"] + (["mfma16d", "mfma16d_w4", "mfma16h", "mfma16h_w4"] if dtype != torch.float32 else [])
    for variant in variants:
        O, L = hip_forward(Q, K, V, causal=causal, variant=variant)
        if dtype == torch.float32:
            assert torch.allclose(O_ref, O, atol=1e-4, rtol=1e-5), variant
        assert (O.float() - O_ref).abs().max() <= O_TOL[dtype], variant
        check_L(L, L_ref, dtype)


def test_concurrent_streams_are_independent():
    """The library is re-entrant and launches on the stream it is given: two problems on two streams at once."""
    torch.manual_seed(5)
    a = [torch.randn(2, 8, 1024, 128, device=DEV).bfloat16() for _ in range(3)]
    b = [torch.randn(1, 8, 2048, 128, device=DEV).bfloat16() for _ in range(3)]
    Oa, La = fa.flash_attention_forward(*a, DEV, causal=True)
    Ob, Lb = fa.flash_attention_forward(*b, DEV, causal=False)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(4):
        with torch.cuda.stream(s1):
            o1 = fa.flash_attention_forward(*a, DEV, causal=True)
        with torch.cuda.stream(s2):
            o2 = fa.flash_attention_forward(*b, DEV, causal=False)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    for (o1, o2) in outs:
        assert torch.equal(o1[0], Oa) and torch.equal(o1[1], La)
        assert torch.equal(o2[0], Ob) and torch.equal(o2[1], Lb)


@pytest.mark.parametrize("causal", [False, True])
def test_fp64_generic(oracle, causal):
    Q, K, V = _rand((1, 2, 80, 32), torch.float64, seed=5)
    O_ref, L_ref = _oracle(oracle, Q, K, V, torch.float64, causal)
    O, L = hip_forward(Q, K, V, causal=causal)
    assert (O - O_ref).abs().max() < 1e-12 and (L - L_ref).abs().max() < 1e-12


@pytest.mark.parametrize("dtype", [torch.float8_e5m2, torch.float8_e4m3fn])
@pytest.mark.parametrize("causal", [False, True])
def test_fp8_generic_vs_oracle(oracle, dtype, causal):
    # P is rounded to fp8 RELATIVE TO THE RUNNING MAX of its key tile (kernels.py:94,98), so with 2-3
    # mantissa bits the result depends on the key-tile size (true of the reference's autotuned tiles too):
    # run the oracle with the generic kernel's 64-key tile.
    Q, K, V = _rand((1, 2, 128, 64), dtype, seed=9, spread=0.5)
    f = lambda t: t.float().numpy()
    O_ref, L_ref = (torch.from_numpy(x) for x in oracle.forward(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal,
                                                                 B_r=16, B_c=64))
    O, L = hip_forward(Q, K, V, causal=causal, variant="generic")
    assert O.dtype == dtype
    O, L = O.float(), L.float()
    step = 0.25 if dtype == torch.float8_e5m2 else 0.125  # one ulp, relative
    assert (O == O_ref).float().mean() > 0.95
    assert ((O - O_ref).abs() <= step * O_ref.abs() + 2 ** -16).all()
    assert (L == L_ref).float().mean() > 0.95


@pytest.mark.parametrize("dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("variant", ["mfma8x", "mfma8x_w4", "auto"])
@pytest.mark.parametrize("shape", [(1, 2, 64, 128), (2, 2, 320, 128), (1, 1, 77, 128), (1, 8, 1024, 128), (1, 3, 191, 128),
                                   (3, 1, 513, 128), (1, 2, 2049, 128)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fp8_mfma_kernel(oracle, dtype, causal, variant, shape):
    """fp8 on the matrix cores (BASELINE.json c5 family).  P is rounded to fp8 relative to the running max of ITS
    32-key block schedule (and a deferred-max threshold), so element-wise equality with a differently tiled run is
    not expected at 2-3 mantissa bits; the bars are statistical, against the exact (fp64) attention of the fp8
    inputs and against the generic kernel (same roundings, 64-key tiles):
        median relative error <= 1 ulp/2 of the format, 99th percentile <= 3 ulp; >= 85 % of elements within one
        fp8 ulp of the generic kernel's output; L within one ulp."""
    Q, K, V = _rand(shape, dtype, seed=11 + shape[2], spread=0.5)
    f = lambda t: t.float().numpy()
    O64, L64 = oracle.sdpa_f64(f(Q), f(K), f(V), causal=causal)
    O, L = hip_forward(Q, K, V, causal=causal, variant=variant)
    Og, Lg = hip_forward(Q, K, V, causal=causal, variant="generic")
    assert O.dtype == dtype and L.dtype == dtype
    O, L, Og, Lg = O.float(), L.float(), Og.float(), Lg.float()
    ulp = 2.0 ** -3 if dtype == torch.float8_e4m3fn else 2.0 ** -2   # relative spacing of the format
    ref = torch.from_numpy(O64).float()
    rel = ((O - ref).abs() / ref.abs().clamp(min=0.05)).flatten()
    assert torch.isfinite(O).all()
    assert rel.median() <= ulp / 2 and rel.kthvalue(int(0.99 * rel.numel())).values <= 3 * ulp
    near = ((O - Og).abs() <= ulp * Og.abs() + 2.0 ** -9).float().mean()
    assert near >= 0.85, near
    Lref = torch.from_numpy(L64).float()
    assert ((L - Lref).abs() <= ulp * Lref.abs() + 1e-3).all()


def test_golden_fp8_e5m2_on_the_gpu(oracle):
    """the reference-held fp8 fixture f8e5m2_seed10 (the reference kernel under the Triton interpreter, 16 x 32 tiles) on the
    GPU.  The chain: fixture = restatement + the interpreter's e5m2 rounding quirk, restatement with true RTNE = C oracle (both
    shown in tests/test_oracle.py::test_fp8_*), C oracle at the kernel's 64-key tile = generic kernel (here, element-wise).
    Against the fixture itself the kernel agrees as often as the oracle at the same tile does (87 % of elements: the quirk and
    the tile width account for the rest) and is closer to the exact attention than the fixture is."""
    g = load_golden("f8e5m2_seed10")
    Q, K, V = (torch.from_numpy(g[k].copy()).view(torch.float8_e5m2) for k in "QKV")
    O_gold = torch.from_numpy(g["O_ref"].copy()).view(torch.float8_e5m2).float()
    f = lambda t: t.float().numpy()
    O_or, L_or = oracle.forward(f(Q), f(K), f(V), "float8_e5m2", B_r=16, B_c=64)
    O, L = hip_forward(Q, K, V, variant="generic")
    O, L, O_or, L_or = O.float(), L.float().flatten(), torch.from_numpy(O_or), torch.from_numpy(L_or).flatten()
    assert (O == O_or).float().mean() >= 0.99, (O == O_or).float().mean()
    assert ((O - O_or).abs() <= 0.25 * O_or.abs() + 2.0 ** -16).all() and (L == L_or).float().mean() >= 0.97
    assert (O == O_gold).float().mean() >= (O_or == O_gold).float().mean() - 0.01
    exact = torch.from_numpy(g["O_sdpa"])
    assert (O - exact).abs().mean() <= (O_gold - exact).abs().mean()


FP8_THR = {torch.float8_e4m3fn: 8.5, torch.float8_e5m2: 15.0}    # deferral threshold of the running maximum (fa2_mfma8x.hip, fa2_a64.hip)


def _fp8_close(O, L, O_ref, L_ref, V, step, same_o=0.99, same_l=0.97, l_abs=1e-3):
    O, L, O_ref, L_ref = O.float(), L.float().flatten(), O_ref.float(), L_ref.float().flatten()
    assert (O == O_ref).float().mean() >= same_o and (L == L_ref).float().mean() >= same_l, ((O == O_ref).float().mean(), (L == L_ref).float().mean())
    assert ((O - O_ref).abs() <= step * O_ref.abs() + 0.5 * step * V.float().abs().max()).all()
    assert ((L - L_ref).abs() <= step * L_ref.abs() + l_abs).all()


@pytest.mark.parametrize("dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
def test_generated_fp8_kernel_a8(oracle, dtype):
    """the generated fp8 kernel (variant a8, asm/fa2_a8_gen.py: the a64 structure on v_mfma_f32_32x32x64_f8f6f4 with P.V on its
    block-scaled form; N a multiple of 256): element by element against the oracle's deferred-maximum mode with an
    INTEGER running maximum (ceil_m: the kernel never rescales O, the power of two rides in the MFMA's scale operand) -- the bar of
    fa2_mfma8x below; against fa2_mfma8x itself (whose maximum is not rounded up: P is rounded against another reference) within
    the same bound; several jobs per workgroup, narrow and N(0, 1) inputs (the maximum moves in most steps), (B, N, H, d)-strided
    inputs, the guard path (a row's maximum jumps by several hundred log2 units) and the table's choice"""
    step = 0.25 if dtype == torch.float8_e5m2 else 0.125
    f = lambda t: t.float().numpy()
    for shape, seed, spread in (((1, 2, 256, 128), 5, 0.5), ((2, 3, 512, 128), 6, 0.5), ((1, 5, 1024, 128), 7, 1.0), ((3, 120, 768, 128), 8, 0.5),
                                ((1, 2, 2048, 128), 9, 1.0)):
        Q, K, V = _rand(shape, dtype, seed=seed, spread=spread)
        O, L = hip_forward(Q, K, V, variant="a8")
        O8, L8 = hip_forward(Q, K, V, variant="mfma8x")
        _fp8_close(O, L, O8, L8, V, step, same_o=0.5, same_l=0.9, l_abs=1.5 * step)    # (two valid roundings of P: 60-85 % of O equal between the oracle's two modes)
        if shape[0] * shape[1] <= 6:
            O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=False, G=32, B_c=64, thr=FP8_THR[dtype],
                                                   sum_rounded=True, ceil_m=True)
            _fp8_close(O, L, torch.from_numpy(O_ref), torch.from_numpy(L_ref), V, step)
    Q, K, V = _rand((1, 2, 512, 128), dtype, seed=10, spread=0.5)
    K[:, :, 500] = (Q[:, :, 7].float() * 8.0).to(dtype)     # row 7's score against key 500: ~ 8 |q|^2 = 256 -> m - m_O far beyond KMAX = 64
    O, L = hip_forward(Q, K, V, variant="a8")
    O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=False, G=32, B_c=64, thr=FP8_THR[dtype], sum_rounded=True, ceil_m=True)
    assert torch.isfinite(O.float()).all()
    _fp8_close(O, L, torch.from_numpy(O_ref), torch.from_numpy(L_ref), V, step)
    u8 = lambda t: t.contiguous().view(torch.uint8)
    gen = torch.Generator().manual_seed(9)
    Q, K, V = ((torch.randn(2, 512, 3, 128, generator=gen) * 0.5).to(dtype).transpose(1, 2) for _ in range(3))     # row stride 3 * 128 bytes
    O, L = hip_forward(Q, K, V, variant="a8")
    O2, L2 = hip_forward(Q.contiguous(), K.contiguous(), V.contiguous(), variant="a8")
    assert torch.equal(u8(O), u8(O2)) and torch.equal(u8(L), u8(L2))
    for shape, seed, causal in (((1, 2, 300, 128), 25, False), ((2, 3, 1000, 128), 26, False), ((1, 4, 2049, 128), 27, True), ((1, 24, 1333, 128), 28, True)):
        Q, K, V = _rand(shape, dtype, seed=seed, spread=0.6)       # ragged N (range-checked rows, masked key tail): canaries behind O and L
        N = shape[2]
        arena_o = torch.full((shape[0], shape[1], N + 8, 128), 1.0, device=DEV).to(dtype)
        arena_l = torch.full((shape[0], shape[1], N + 8, 1), 1.0, device=DEV).to(dtype)
        Oa, La = arena_o[:, :, :N], arena_l[:, :, :N]
        _lib.fa2_fwd(Q.to(DEV), K.to(DEV), V.to(DEV), Oa, La, fa.convert_triton_dtype(dtype), causal=causal, variant=_lib.VARIANT_A8)
        torch.cuda.synchronize()
        assert (arena_o[:, :, N:].float() == 1.0).all() and (arena_l[:, :, N:].float() == 1.0).all(), shape
        O8, L8 = hip_forward(Q, K, V, causal=causal, variant="mfma8x")
        _fp8_close(Oa.cpu(), La.cpu(), O8, L8, V, step, same_o=0.5, same_l=0.9, l_abs=1.5 * step)
        if shape[0] * shape[1] <= 6:
            O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, G=32, B_c=64, thr=FP8_THR[dtype],
                                                   sum_rounded=True, ceil_m=True)
            _fp8_close(Oa.cpu(), La.cpu(), torch.from_numpy(O_ref), torch.from_numpy(L_ref), V, step, same_o=0.98)
    x = torch.zeros(1, 2, 192, 128).to(dtype)
    with pytest.raises(TypeError):
        hip_forward(x, x, x, variant="a8")                 # N below 256
    for shape, seed, spread in (((1, 2, 256, 128), 15, 0.5), ((2, 3, 768, 128), 16, 0.7), ((1, 24, 1024, 128), 17, 1.0), ((1, 4, 2048, 128), 18, 1.0),
                                ((1, 8, 4096, 128), 19, 1.0), ((1, 8, 4352, 128), 20, 0.7)):      # (16 / 17 query blocks: the last with, the first without the downward walk)
        Q, K, V = _rand(shape, dtype, seed=seed, spread=spread)      # the causal form (light jobs walk downwards from N = 512 on)
        O, L = hip_forward(Q, K, V, causal=True, variant="a8")
        O8, L8 = hip_forward(Q, K, V, causal=True, variant="mfma8x")
        # (two roundings of P: l = sum of the rounded P is off by up to half a step relatively, log2 l by 0.72 steps absolutely --
        # which shows where |L| is small, in a causal problem's first rows)
        _fp8_close(O, L, O8, L8, V, step, same_o=0.5, same_l=0.9, l_abs=1.5 * step)
        if shape[0] * shape[1] <= 6:
            O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=True, G=32, B_c=64, thr=FP8_THR[dtype],
                                                   sum_rounded=True, ceil_m=True)
            # (N(0, 1) inputs at N = 2048: 98.6 % of the e4m3 bytes equal, every one inside the bound -- near-one-hot rows turn a last-bit
            # difference of v_exp_f32 against exp2f into a rounding step of O more often)
            _fp8_close(O, L, torch.from_numpy(O_ref), torch.from_numpy(L_ref), V, step, same_o=0.98)
    en = fa.convert_triton_dtype(dtype)
    # the table's choice on large even grids (pick_variant() in csrc/fa2_api.hip has the numbers); 1.5 jobs per CU and small grids stay
    assert _lib.query_tile(16384, 128, en, False, B=16, H=8)[0] == _lib.VARIANT_A8
    assert _lib.query_tile(2048, 128, en, False, B=1, H=48)[0] == _lib.VARIANT_MFMA8X_W4
    assert _lib.query_tile(2048, 128, en, False, B=1, H=16)[0] in (_lib.VARIANT_MFMA8X, _lib.VARIANT_MFMA8X_W4)
    assert _lib.query_tile(16384, 128, en, True, B=16, H=8)[0] == _lib.VARIANT_A8
    assert _lib.query_tile(4096, 128, en, True, B=1, H=16)[0] == _lib.VARIANT_MFMA8X_W4      # (half a unit per CU)


@pytest.mark.parametrize("dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("variant", ["mfma8x", "mfma8x_w4"])
def test_fp8_mfma_kernel_element_wise_against_the_oracle_in_deferred_maximum_mode(oracle, dtype, causal, variant):
    """fa2_mfma8x keeps the running maximum of a wave's 32 rows while none of them exceeds it by 8.5 (e4m3) / 15 (e5m2) log2 units
    within a 64-key unit, forms exp2(fma(S, c, -m)) and sums the rounded P on the matrix pipe: exactly oracle.forward_deferred(G=32,
    B_c=64, thr).  Against that restatement, element by element: >= 99 % of O bit-identical (a systematic one-ulp bias would fail
    this bar); an element that differs is off by its own rounding step plus at most one rounding step of P times max |V| (a P
    that v_exp_f32's last bit rounds the other way moves O by its share p / l of V)."""
    for shape, seed in (((2, 2, 320, 128), 5), ((1, 3, 191, 128), 6), ((1, 2, 1024, 128), 7)):
        Q, K, V = _rand(shape, dtype, seed=seed, spread=0.5)
        f = lambda t: t.float().numpy()
        O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, G=32, B_c=64, thr=FP8_THR[dtype],
                                               sum_rounded=True)
        O, L = hip_forward(Q, K, V, causal=causal, variant=variant)
        O, L, O_ref, L_ref = O.float(), L.float().flatten(), torch.from_numpy(O_ref), torch.from_numpy(L_ref).flatten()
        step = 0.25 if dtype == torch.float8_e5m2 else 0.125   # one ulp, relative
        same = (O == O_ref).float().mean().item()
        assert same >= 0.99, (shape, same)
        bound = step * O_ref.abs() + 0.5 * step * V.float().abs().max()
        assert ((O - O_ref).abs() <= bound).all(), (shape, ((O - O_ref).abs() / bound).max().item())
        assert (L == L_ref).float().mean() >= 0.97 and ((L - L_ref).abs() <= step * L_ref.abs() + 1e-3).all()


@pytest.mark.parametrize("dtype", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("variant", ["mfma8x", "mfma8x_w4"])
@pytest.mark.parametrize("causal", [False, True])
def test_fp8_rescale_and_layouts(oracle, dtype, variant, causal):
    """fp8 matrix kernels where the running max jumps late (keys grow in norm, one spiked key in the last unit: the
    rescale of O and of the row sum -- kept in MFMA accumulators by mfma8x -- is taken after O is non-zero), with
    (B, N, H, d)-strided Q/K/V views as the reference's wrappers may pass (src/test_kernels.py strided case).
    Bars as in test_fp8_mfma_kernel, against the exact attention of the fp8 inputs."""
    B, H, N, d = 2, 3, 448, 128
    gen = torch.Generator().manual_seed(77)
    Q, K, V = ((torch.randn(B, N, H, d, generator=gen) * 0.4) for _ in range(3))
    K = K * torch.linspace(0.3, 1.6, N).view(1, N, 1, 1)
    K[:, N - 3] = Q[:, 11] * 2.0  # row 11's maximum jumps in the final 64-key unit
    Q, K, V = (t.to(dtype).transpose(1, 2) for t in (Q, K, V))  # (B, H, N, d) views of (B, N, H, d) storage
    assert not Q.is_contiguous()
    f = lambda t: t.float().contiguous().numpy()
    O64, L64 = oracle.sdpa_f64(f(Q), f(K), f(V), causal=causal)
    O, L = fa.flash_attention_forward(Q.to(DEV), K.to(DEV), V.to(DEV), DEV, causal=causal, variant=variant)
    torch.cuda.synchronize()
    O, L = O.cpu().float(), L.cpu().float()
    ulp = 2.0 ** -3 if dtype == torch.float8_e4m3fn else 2.0 ** -2
    ref = torch.from_numpy(O64).float()
    rel = ((O - ref).abs() / ref.abs().clamp(min=0.05)).flatten()
    assert torch.isfinite(O).all()
    assert rel.median() <= ulp / 2 and rel.kthvalue(int(0.99 * rel.numel())).values <= 3 * ulp
    Lref = torch.from_numpy(L64).float()
    assert ((L - Lref).abs() <= ulp * Lref.abs() + 1e-3).all()


def test_fp8_c5_head_shard_properties():
    """BASELINE.json configs[4] (B16 H64 N16384 d128 fp8, head/batch-sharded over 8 GPUs) at full N on a slice of one
    GPU's shard, through size-independent properties: run-to-run bit equality, bit-identical head indexing (head
    shards run separately == the full run), V = 1 => O = 1 exactly, key permutation invariance within fp8 rounding."""
    dtype, B, H, N, d = torch.float8_e4m3fn, 1, 8, 16384, 128
    torch.manual_seed(42)
    Q, K, V = ((torch.randn(B, H, N, d, device=DEV) * 0.5).to(dtype) for _ in range(3))
    # the kernel the table picks for a GPU's whole shard (B16 H8: 8 192 jobs) -- pinned, because this slice's two-head pieces are
    # grids of their own (128 jobs) for which the table would pick another kernel: equal bits need the same kernel
    assert _lib.query_tile(N, d, _lib.FA2_DTYPE_F8E4M3, False, B=16, H=8)[0] == _lib.VARIANT_A8
    run = lambda q, k, v: fa.flash_attention_forward(q, k, v, DEV, variant="a8")
    O, L = run(Q, K, V)
    O2, L2 = run(Q, K, V)
    assert torch.equal(O.view(torch.uint8), O2.view(torch.uint8)) and torch.equal(L.view(torch.uint8), L2.view(torch.uint8))
    parts = [run(Q[:, h0:h0 + 2].contiguous(), K[:, h0:h0 + 2].contiguous(), V[:, h0:h0 + 2].contiguous())[0] for h0 in range(0, H, 2)]
    assert torch.equal(torch.cat(parts, dim=1).view(torch.uint8), O.view(torch.uint8))
    O1, _ = run(Q, K, torch.ones(B, H, N, d, device=DEV).to(dtype))
    assert (O1.float() == 1).all()
    assert torch.isfinite(O.float()).all() and torch.isfinite(L.float()).all()
    # one (b, h) against the exact attention of the fp8 inputs (fp64 on the GPU), statistical bars of the fp8 tests
    S = Q[0, 0].double() @ K[0, 0].double().T
    ref = (torch.softmax(S, dim=-1) @ V[0, 0].double()).float()
    rel = ((O[0, 0].float() - ref).abs() / ref.abs().clamp(min=0.05)).flatten()
    assert rel.median() <= 2.0 ** -4 and rel.kthvalue(int(0.99 * rel.numel())).values <= 3 * 2.0 ** -3
    lse2 = (torch.logsumexp(S, dim=-1) * math.log2(math.e)).float()
    assert ((L[0, 0].float().flatten() - lse2).abs() <= 2.0 ** -3 * lse2.abs() + 1e-3).all()


@pytest.mark.parametrize("dtype,variant", [(torch.float32, "mfma32"), (torch.bfloat16, "mfma16"),
                                           (torch.bfloat16, "mfma16_w8"), (torch.float32, "generic"),
                                           (torch.bfloat16, "mfma16d"), (torch.bfloat16, "mfma16h"),
                                           (torch.float16, "mfma16h_w4")])
def test_scale_extension(oracle, dtype, variant):
    Q, K, V = _rand((1, 2, 192, 128), dtype, seed=21)
    scale = 1.0 / math.sqrt(128)
    O_ref, L_ref = _oracle(oracle, Q, K, V, dtype, True, scale)
    O, L = hip_forward(Q, K, V, causal=True, scale=scale, variant=variant)
    assert (O.float() - O_ref).abs().max() <= O_TOL[dtype]
    check_L(L, L_ref, dtype)


@pytest.mark.parametrize("dtype,variant", [(torch.float32, "mfma32"), (torch.bfloat16, "mfma16"),
                                           (torch.float16, "mfma16_w8"), (torch.float32, "generic"),
                                           (torch.bfloat16, "mfma16d"), (torch.float16, "mfma16d"),
                                           (torch.bfloat16, "mfma16h"),
                                           (torch.float16, "mfma16h"), (torch.bfloat16, "mfma16h_w4")])
def test_rescale_branch_is_exercised(oracle, dtype, variant):
    """Online-softmax rescaling: make the running max jump at LATE tiles (keys grow in norm and one
    spiked key sits in the last tile), so O *= coeff is taken with coeff << 1 after O is non-zero."""
    B, H, N, d = 1, 2, 512, 128
    Q, K, V = _rand((B, H, N, d), torch.float32, seed=33, spread=0.3)
    K = K * torch.linspace(0.2, 2.0, N).view(1, 1, N, 1)
    K[:, :, N - 5] = Q[:, :, 7] * 8.0   # row 7's max jumps by ~130 log2 units in the final tile (thresholds: 12 / 60)
    Q, K, V = (t.to(dtype) for t in (Q, K, V))
    O_ref, L_ref = _oracle(oracle, Q, K, V, dtype, False)
    O, L = hip_forward(Q, K, V, variant=variant)
    assert (O.float() - O_ref).abs().max() <= O_TOL[dtype]
    check_L(L, L_ref, dtype)


def test_no_out_of_bounds_writes_for_ragged_N():
    """O and L of a ragged problem sit inside a poisoned arena; nothing outside them may change."""
    for dtype, variant, d in ((torch.float32, "mfma32", 64), (torch.bfloat16, "mfma16", 64),
                              (torch.bfloat16, "mfma16_w8", 64), (torch.float32, "generic", 64),
                              (torch.bfloat16, "mfma16d", 128),
                              (torch.bfloat16, "mfma16h", 128), (torch.float16, "mfma16h_w4", 64),
                              (torch.bfloat16, "mfma16k", 128), (torch.float16, "mfma16k", 64),
                              (torch.float8_e4m3fn, "mfma8x", 128), (torch.float8_e5m2, "mfma8x_w4", 128),
                              (torch.float8_e5m2, "mfma8x", 128)):
        B, H, N = 1, 2, 77
        Q, K, V = (t.to(DEV) for t in _rand((B, H, N, d), dtype, seed=3))
        arena = torch.full((3, B, H, N, d), 7.0, device=DEV).to(dtype)  # 7 is exact in every dtype here, fp8 included
        arena_l = torch.full((3, B, H, N, 1), 7.0, device=DEV).to(dtype)
        O, L = arena[1], arena_l[1]
        _lib.fa2_fwd(Q, K, V, O, L, fa.convert_triton_dtype(dtype), variant=_lib.VARIANTS[variant])
        torch.cuda.synchronize()
        assert all((t.float() == 7).all() for t in (arena[0], arena[2], arena_l[0], arena_l[2]))
        assert torch.isfinite(O.float()).all() and (O.float().abs() < 7).all()


def test_hip_graph_capture_and_replay():
    """The launch path has no allocation and no synchronisation (contract table): forward and backward
    can be captured into a HIP graph after one warm-up call (the first call of a kernel family sets its LDS attribute)
    and replayed on new contents of the same buffers, bit-identical to eager launches."""
    from flash_attention_dlrs_amd import flash_attention_backward
    for dtype, shape, causal in ((torch.bfloat16, (2, 4, 512, 128), True), (torch.float8_e4m3fn, (1, 4, 320, 128), False),
                                 (torch.float32, (1, 2, 200, 64), False)):
        gen = torch.Generator().manual_seed(5)
        mk = lambda: (torch.randn(*shape, generator=gen) * 0.5).to(dtype).to(DEV)
        Q, K, V, dO = mk(), mk(), mk(), mk()
        bwd = dtype != torch.float8_e4m3fn  # the backward rejects fp8
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on the capture stream
            O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
            if bwd:
                flash_attention_backward(Q, K, V, O, dO, L, DEV, causal=causal)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            O_g, L_g = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
            grads_g = flash_attention_backward(Q, K, V, O_g, dO, L_g, DEV, causal=causal) if bwd else ()
        for t in (Q, K, V, dO):
            t.copy_(mk())
        g.replay()
        torch.cuda.synchronize()
        O_e, L_e = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
        raw = lambda t: t.view(torch.uint8) if t.element_size() == 1 else t
        assert torch.equal(raw(O_g), raw(O_e)) and torch.equal(raw(L_g), raw(L_e))
        if bwd:
            grads_e = flash_attention_backward(Q, K, V, O_e, dO, L_e, DEV, causal=causal)
            assert all(torch.equal(a, b) for a, b in zip(grads_g, grads_e))


def test_unsupported_variant_and_dtype_errors():
    x = torch.zeros(1, 1, 32, 32, device=DEV)
    with pytest.raises(TypeError):
        fa.flash_attention_forward(x.half(), x.half(), x.half(), DEV, variant="mfma32")   # an fp32 kernel
    with pytest.raises(TypeError):
        fa.FlashAttention.apply(x.int(), x.int(), x.int())
    with pytest.raises(ValueError):
        fa.FlashAttention.apply(x, x[:, :, :16], x)
    # experimental kernels are not in the product library (include/fa2_fwd.h lists what is): the product surface does not know
    # their names, and the library rejects their ids
    y = torch.zeros(1, 1, 64, 128, device=DEV, dtype=torch.bfloat16)
    for name in ("mfma16p", "mfma16x", "mfma16s", "mfma8", "mfma8u", "abl_noexp"):
        assert name not in _lib.VARIANTS and name in _lib.EXPERIMENTAL_VARIANTS
        with pytest.raises(KeyError):
            fa.flash_attention_forward(y, y, y, DEV, variant=name)
        O, L = torch.empty_like(y), torch.empty(1, 1, 64, 1, device=DEV, dtype=torch.bfloat16)
        with pytest.raises(ValueError):
            _lib.fa2_fwd(y, y, y, O, L, _lib.FA2_DTYPE_BF16, variant=_lib.EXPERIMENTAL_VARIANTS[name])


def test_autograd_surface_backward_runs():
    torch.manual_seed(0)
    Q, K, V = (torch.randn(1, 2, 64, 64, device=DEV, requires_grad=True) for _ in range(3))
    O = fa.FlashAttention.apply(Q, K, V)
    O_t = torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1)
    dO = torch.randn_like(O)
    g = torch.autograd.grad(O, (Q, K, V), dO)
    g_t = torch.autograd.grad(O_t, (Q, K, V), dO)
    # tolerances of the reference's backward check (test_correctness.py:60-62)
    for a, b, atol in zip(g, g_t, (9e-4, 7e-4, 7e-5)):
        assert torch.allclose(b, a, atol=atol, rtol=1e-5)


# ----------------------------------------------------------------------------- (c) full sizes
def sdpa_ref(Q, K, V, causal):
    """fp32 SDPA(scale=1) of the already-rounded inputs on the GPU, one batch element at a time."""
    outs = []
    for b in range(Q.shape[0]):
        with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.MATH):
            outs.append(torch.nn.functional.scaled_dot_product_attention(
                Q[b:b + 1].float(), K[b:b + 1].float(), V[b:b + 1].float(), scale=1.0, is_causal=causal))
    return torch.cat(outs)


CONFIGS = {  # BASELINE.json configs[1], configs[2]
    "c2": dict(shape=(2, 8, 1024, 64), dtype=torch.float16, causal=False),
    "c3": dict(shape=(4, 32, 4096, 128), dtype=torch.bfloat16, causal=True),
}


@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_full_size_vs_live_sdpa_and_properties(cfg):
    c = CONFIGS[cfg]
    dtype, causal = c["dtype"], c["causal"]
    torch.manual_seed(42)  # src/bench.py:26
    Q, K, V = (torch.randn(*c["shape"], device=DEV).to(dtype) for _ in range(3))
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
    ref = sdpa_ref(Q, K, V, causal)
    tol = O_TOL[dtype]
    assert (O.float() - ref).abs().max().item() <= tol
    # determinism + bit-identical head indexing: running head shards separately == the full run
    O2, L2 = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
    assert torch.equal(O, O2) and torch.equal(L, L2)
    H = Q.shape[1]
    parts = [fa.flash_attention_forward(Q[:, h0:h0 + H // 4].contiguous(), K[:, h0:h0 + H // 4].contiguous(),
                                        V[:, h0:h0 + H // 4].contiguous(), DEV, causal=causal)[0]
             for h0 in range(0, H, H // 4)]
    assert torch.equal(torch.cat(parts, dim=1), O)
    # V = 1  =>  O = 1 (rows of softmax sum to one), up to the rounding of P and O
    ones = torch.ones_like(V)
    O1, _ = fa.flash_attention_forward(Q, K, ones, DEV, causal=causal)
    assert (O1.float() - 1).abs().max().item() <= tol
    # linearity in V
    V2 = torch.randn_like(V)
    Oa, _ = fa.flash_attention_forward(Q, K, V2, DEV, causal=causal)
    Os, _ = fa.flash_attention_forward(Q, K, (V.float() + V2.float()).to(dtype), DEV, causal=causal)
    assert (Os.float() - (O.float() + Oa.float())).abs().max().item() <= 3 * tol
    # L is the log2-domain log-sum-exp of the scores (checked on one (b, h))
    S = Q[0, 0].float() @ K[0, 0].float().T
    if causal:
        S = S.masked_fill(~torch.ones_like(S, dtype=torch.bool).tril(), float("-inf"))
    lse2 = torch.logsumexp(S, dim=-1) * math.log2(math.e)
    check_L(L[0, 0].cpu(), lse2.cpu(), dtype)
    if not causal:  # permuting the keys (and values) leaves O unchanged
        perm = torch.randperm(Q.shape[2], device=DEV)
        Op, _ = fa.flash_attention_forward(Q, K[:, :, perm].contiguous(), V[:, :, perm].contiguous(), DEV)
        assert (Op.float() - O.float()).abs().max().item() <= 2 * tol


def test_d_inv_quarter_scaled_inputs_c3_shape():
    """scale=1 on N(0,1) data makes softmax nearly one-hot at d=128; also check inputs scaled by d^-1/4
    (equivalent to the usual 1/sqrt(d)), where many keys contribute to every row."""
    torch.manual_seed(1)
    shape = (1, 8, 4096, 128)
    Q, K, V = (torch.randn(*shape, device=DEV) for _ in range(3))
    Q, K = (Q * 128 ** -0.25).bfloat16(), (K * 128 ** -0.25).bfloat16()
    V = V.bfloat16()
    for causal in (False, True):
        O, _ = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
        assert (O.float() - sdpa_ref(Q, K, V, causal)).abs().max().item() <= 1e-2


def test_tuned_variant_falls_back_when_it_cannot_run_the_problem(tmp_path):
    """The tuner's key does not see strides or alignment: a table entry naming a kernel that needs unit d-stride must not turn
    a call the reference accepts (arbitrary strides) into an error -- the static table takes over for that call."""
    from flash_attention_dlrs_amd import autotune
    path = str(tmp_path / "tile_table.json")
    Qc = torch.zeros(1, 8, 512, 128, device=DEV, dtype=torch.bfloat16)
    json.dump({autotune.key_of(Qc, True): {"variant": "a64", "ms": {}, "table_id": autotune.table_id(DEV)}}, open(path, "w"))
    autotune.enable(True, path)
    try:
        gen = torch.Generator().manual_seed(5)
        Q, K, V = (torch.randn(1, 8, 512, 256, generator=gen).bfloat16().to(DEV)[..., ::2] for _ in range(3))  # d-stride 2
        assert Q.stride(3) == 2 and autotune.key_of(Q, True) == autotune.key_of(Qc, True)
        O, _ = fa.flash_attention_forward(Q, K, V, DEV, causal=True)
        ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0, is_causal=True)
        assert (O.float() - ref).abs().max().item() <= O_TOL[torch.bfloat16]
        Oc, _ = fa.flash_attention_forward(Q.contiguous(), K.contiguous(), V.contiguous(), DEV, causal=True)   # the tuned kernel
        assert (Oc.float() - ref).abs().max().item() <= O_TOL[torch.bfloat16]
    finally:
        autotune.enable(False)


def test_runtime_tuner_picks_a_parity_tested_variant(tmp_path, oracle):
    """with the on-box tuner enabled the first call measures the candidates, persists the
    choice, and the result stays within the documented tolerance whichever variant wins."""
    from flash_attention_dlrs_amd import autotune
    Q, K, V = _rand((1, 8, 512, 128), torch.bfloat16, seed=77)
    O_ref, _ = _oracle(oracle, Q, K, V, torch.bfloat16, True)
    path = str(tmp_path / "tile_table.json")
    autotune.enable(True, path)
    try:
        O, _ = fa.flash_attention_forward(Q.to(DEV), K.to(DEV), V.to(DEV), DEV, causal=True)
        table = json.load(open(path))
        (key, entry), = table.items()
        assert key == "bfloat16:d128:N512:causal:small" and entry["variant"] in _lib.VARIANTS and "auto" in entry["ms"]
        O2, _ = fa.flash_attention_forward(Q.to(DEV), K.to(DEV), V.to(DEV), DEV, causal=True)  # served from the table
        assert torch.equal(O, O2)
    finally:
        autotune.enable(False)
    assert (O.float().cpu() - O_ref).abs().max() <= O_TOL[torch.bfloat16]
