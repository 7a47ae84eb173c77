"""GPU parity tests of the generated assembly kernel (variant "a64": f16 / bf16, d = 128, N a multiple of 256), through the C ABI.

(a) seeded inputs against the CPU oracle (oracle/fa2_oracle.c) at sizes it finishes in seconds: every (dtype, causal, scale),
    several jobs per workgroup, permuted (B, N, H, d) storage, padded row strides;
(b) the deferred-rescale path forced by a late jump of one row's maximum; NaN / Inf inputs against the oracle;
(c) BASELINE.json configs[3] and configs[4] at their full per-GPU shard sizes (c4: B8 H8 N8192 bf16 -- the default table
    sends it to this kernel; c5: B16 H8 N16384 e4m3 -- the fp8 kernel), through size-independent properties.
Tolerances as tests/test_fwd_parity.py: bf16 |O - ref| <= 5e-2, f16 6e-3, L one ulp of the dtype.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import flash_attention_dlrs_amd as fa  # noqa: E402
from flash_attention_dlrs_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
O_TOL = {torch.float16: 6e-3, torch.bfloat16: 5e-2}
ORACLE_NAME = {torch.float16: "float16", torch.bfloat16: "bfloat16"}


def ulp(dtype, x):
    mant = {torch.float16: 10, torch.bfloat16: 7}[dtype]
    return 2.0 ** (math.floor(math.log2(max(abs(x), 1e-30))) - mant)


def rand3(shape, dtype, seed, spread=1.0):
    gen = torch.Generator().manual_seed(seed)
    return tuple((torch.randn(*shape, generator=gen) * spread).to(dtype) for _ in range(3))


def oracle_fwd(oracle, Q, K, V, dtype, causal, scale=1.0):
    f = lambda t: t.float().contiguous().numpy()
    O, L = oracle.forward(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, scale=scale, B_r=64, B_c=64)
    return torch.from_numpy(O), torch.from_numpy(L)


def a64(Q, K, V, causal=False, scale=1.0, variant="a64"):
    O, L = fa.flash_attention_forward(Q.to(DEV), K.to(DEV), V.to(DEV), DEV, causal=causal, scale=scale, variant=variant)
    torch.cuda.synchronize()
    return O.cpu(), L.cpu()


def check(O, L, O_ref, L_ref, dtype):
    assert torch.isfinite(O.float()).all()
    assert (O.float() - O_ref).abs().max() <= O_TOL[dtype]
    assert (L.float().flatten() - L_ref.flatten()).abs().max() <= 1.01 * ulp(dtype, L_ref.abs().max().item())


SHAPES = [(1, 1, 256), (2, 3, 512), (1, 2, 768), (1, 5, 1024)]
# the generated kernel in its two matrix shapes: "a64" on v_mfma_f32_32x32x16, "a16" on v_mfma_f32_16x16x32 (same structure, same
# arithmetic order: the tests below hold for both)
GEN_VARIANTS = ["a64", "a16"]
both = pytest.mark.parametrize("variant", GEN_VARIANTS)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
@both
def test_seeded_vs_oracle(oracle, dtype, causal, shape, variant):
    B, H, N = shape
    Q, K, V = rand3((B, H, N, 128), dtype, seed=N + H)
    check(*a64(Q, K, V, causal, variant=variant), *oracle_fwd(oracle, Q, K, V, dtype, causal), dtype)


@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 128 ** -0.5), (torch.float16, 0.25)])
@both
def test_scale_and_default_table(oracle, dtype, scale, variant):
    Q, K, V = rand3((1, 2, 512, 128), dtype, seed=5)
    O_ref, L_ref = oracle_fwd(oracle, Q, K, V, dtype, True, scale)
    check(*a64(Q, K, V, True, scale, variant=variant), O_ref, L_ref, dtype)


def test_default_table_picks_a64_for_the_north_star_shape():
    assert _lib.query_tile(4096, 128, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A64
    assert _lib.query_tile(8192, 128, _lib.FA2_DTYPE_BF16, False)[0] == _lib.VARIANT_A16   # long non-causal bf16: the 16x16x32 form
    assert _lib.query_tile(8192, 128, _lib.FA2_DTYPE_F16, False)[0] == _lib.VARIANT_A64
    assert _lib.query_tile(8192, 128, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A16 and _lib.query_tile(4096, 128, _lib.FA2_DTYPE_BF16, False)[0] == _lib.VARIANT_A16
    assert _lib.query_tile(4000, 128, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A64   # N not a multiple of 256: the ragged kernels
    assert _lib.query_tile(4096, 64, _lib.FA2_DTYPE_BF16, True)[0] != _lib.VARIANT_A64    # d = 64


@both
def test_many_jobs_per_workgroup_and_job_order(oracle, variant):
    """more jobs than CUs: every workgroup walks several jobs (seam: next-job prefetch, epilogue between jobs), causal pairs"""
    dtype = torch.bfloat16
    Q, K, V = rand3((3, 48, 512, 128), dtype, seed=9)      # 288 (b, h) x 2 query blocks = 576 jobs, B * H not a power of two
    for causal in (False, True):
        O, L = a64(Q, K, V, causal, variant=variant)
        # the oracle on a few heads (it takes seconds per head at this size), all heads against each other through SDPA
        for (b, h) in ((0, 0), (1, 17), (2, 47)):
            O_ref, L_ref = oracle_fwd(oracle, Q[b:b + 1, h:h + 1], K[b:b + 1, h:h + 1], V[b:b + 1, h:h + 1], dtype, causal)
            check(O[b:b + 1, h:h + 1], L[b:b + 1, h:h + 1], O_ref, L_ref, dtype)
        ref = torch.nn.functional.scaled_dot_product_attention(Q.to(DEV).float(), K.to(DEV).float(), V.to(DEV).float(),
                                                               scale=1.0, is_causal=causal).cpu()
        assert (O.float() - ref).abs().max() <= O_TOL[dtype]


@both
def test_many_jobs_f16_rescale_across_job_seams(variant):
    """f16 defers the running maximum by at most 15.875 log2 units, so O and l are rescaled often: with several jobs per workgroup
    every rare path (firing, deferred rescale, epilogue, next job's prefetch) meets every other"""
    dtype = torch.float16
    Q, K, V = rand3((2, 160, 512, 128), dtype, seed=13)      # 640 jobs on 256 workgroups
    for causal in (False, True):
        O, _ = a64(Q, K, V, causal, variant=variant)
        ref = torch.nn.functional.scaled_dot_product_attention(Q.to(DEV).float(), K.to(DEV).float(), V.to(DEV).float(),
                                                               scale=1.0, is_causal=causal).cpu()
        assert torch.isfinite(O.float()).all()
        assert (O.float() - ref).abs().max() <= O_TOL[dtype]


@both
def test_strided_inputs(oracle, variant):
    """(B, N, H, d) storage viewed as (B, H, N, d) (row stride H * d), and rows padded to 136 elements"""
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(21)
    Q, K, V = (torch.randn(2, 512, 3, 128, generator=gen).to(dtype).transpose(1, 2) for _ in range(3))
    assert not Q.is_contiguous()
    check(*a64(Q, K, V, True, variant=variant), *oracle_fwd(oracle, Q, K, V, dtype, True), dtype)
    Qp, Kp, Vp = (torch.randn(1, 2, 256, 136, generator=gen).to(dtype)[..., :128] for _ in range(3))
    assert Qp.stride(2) == 136
    O, L = fa.flash_attention_forward(Qp.to(DEV), Kp.to(DEV), Vp.to(DEV), DEV, variant=variant)
    check(O.cpu(), L.cpu(), *oracle_fwd(oracle, Qp, Kp, Vp, dtype, False), dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@both
def test_element_wise_against_the_oracle_in_deferred_maximum_mode(oracle, dtype, causal, variant):
    """the kernel's liberties are exactly the ones oracle.forward_deferred states (running maximum kept per 32-row query block
    while no row of it exceeds m by 60 / 15.875 log2 units, exp2(fma), row sums of the rounded P): against THAT restatement the
    result is compared element by element -- at least 99 % of O bit-identical; an element that differs is off by its own
    rounding step plus at most one rounding step of P times max |V| (a P that v_exp_f32's last bit rounds the other way moves O
    by its share p / l of V -- early causal rows have few keys and large shares).  What remains besides is the fp32 summation
    order of the matrix pipe."""
    B, H, N = 1, 2, 512
    Q, K, V = rand3((B, H, N, 128), dtype, seed=77)
    K[:, :, 300] = (Q[:, :, 200].float() * 0.45).to(dtype)     # one row's maximum jumps past the f16 threshold mid-way
    O, L = a64(Q, K, V, causal, variant=variant)
    f = lambda t: t.float().numpy()
    O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, G=32, B_c=64,
                                           thr=60.0 if dtype == torch.bfloat16 else 15.875, sum_rounded=True)
    O, O_ref = O.float(), torch.from_numpy(O_ref)
    same = (O == O_ref).float().mean().item()
    assert same >= 0.99, same
    mant = 7 if dtype == torch.bfloat16 else 10
    one_ulp = torch.exp2(torch.floor(torch.log2(O_ref.abs().clamp(min=2.0 ** -14))) - mant)
    bound = one_ulp + 2.0 ** -(mant + 1) * V.float().abs().max()
    assert ((O - O_ref).abs() <= bound).all(), ((O - O_ref).abs() / bound).max().item()
    Lf, L_ref = L.float().flatten(), torch.from_numpy(L_ref).flatten()
    assert (Lf == L_ref).float().mean() >= 0.97 and (Lf - L_ref).abs().max() <= 1.01 * ulp(dtype, L_ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@both
def test_rescale_branch_is_exercised(oracle, dtype, variant):
    """the running maximum of one row jumps far beyond the deferral threshold (60 / 15.875 log2 units) in the last tiles, after O
    and l are non-zero: O *= coeff, l *= coeff are taken (cdna_hip_programming.md rule 26)"""
    B, H, N = 1, 2, 1024
    Q, K, V = rand3((B, H, N, 128), torch.float32, seed=33, spread=0.3)
    K = K * torch.linspace(0.2, 2.0, N).view(1, 1, N, 1)
    K[:, :, N - 5] = Q[:, :, 7] * 8.0
    K[:, :, 300] = Q[:, :, 200] * 6.0
    Q, K, V = (t.to(dtype) for t in (Q, K, V))
    for causal in (False, True):
        check(*a64(Q, K, V, causal, variant=variant), *oracle_fwd(oracle, Q, K, V, dtype, causal), dtype)


@both
def test_large_magnitude_V_stays_finite(oracle, variant):
    """bf16 defers the running maximum by up to 60 log2 units, so P may reach 2^60 before a rescale and the fp32 O
    accumulator holds sum(P |V|): finite while |V| stays below about 2^68 / N (3e16 at N = 4096) -- the documented bound
    (the reference, with P <= 1, takes any bf16 V).  |V| ~ 1e15 with a row maximum that jumps late, against the
    plain oracle (relative to the scale of V)"""
    dtype, scale_v = torch.bfloat16, 1e15
    Q, K, V = rand3((1, 2, 1024, 128), torch.float32, seed=41, spread=0.6)
    K[:, :, 900] = Q[:, :, 100] * 3.0          # row 100's maximum moves by ~60 log2 units in the last tiles
    V = V * scale_v
    Q, K, V = (t.to(dtype) for t in (Q, K, V))
    for causal in (False, True):
        O, L = a64(Q, K, V, causal, variant=variant)
        O_ref, L_ref = oracle_fwd(oracle, Q, K, V, dtype, causal)
        assert torch.isfinite(O.float()).all()
        assert ((O.float() - O_ref) / scale_v).abs().max() <= O_TOL[dtype]
        assert (L.float().flatten() - L_ref.flatten()).abs().max() <= 1.01 * ulp(dtype, L_ref.abs().max().item())


@pytest.mark.parametrize("variant", ["a64", "a16", "mfma16h", "mfma16"])
def test_tensors_that_straddle_a_4_gib_address_boundary(variant):
    """every 64-bit address the kernels form (descriptor base + b * stride_b + h * stride_h) must carry out of its low word:
    Q, K, V, O and L are placed so that each crosses a 4-GiB-aligned device address inside its own arena.  (The a64 seam once
    lost exactly that carry: a DMA set-up's s_add_u32 sat between the s_add_u32 / s_addc_u32 of the next job's Q descriptor.)"""
    dtype = torch.bfloat16
    B, H, N, d = 2, 80, 512, 128     # 320 jobs of 256 rows on 256 workgroups: job seams (the next job's descriptors) included
    gen = torch.Generator().manual_seed(5)
    src = [torch.randn(B, H, N, d, generator=gen).to(dtype) for _ in range(3)]
    GiB = 1 << 30

    def straddling(shape, dt):
        nbytes = math.prod(shape) * torch.empty((), dtype=dt).element_size()
        arena = torch.empty(4 * GiB + 2 * nbytes + 4096, dtype=torch.uint8, device=DEV)
        edge = (arena.data_ptr() + nbytes + 4 * GiB - 1) // (4 * GiB) * (4 * GiB)      # first 4-GiB multiple past ptr + nbytes
        start = edge - nbytes // 2 - (edge - nbytes // 2) % 256                          # the tensor's middle sits on the edge
        off = start - arena.data_ptr()
        assert 0 <= off and off + nbytes <= arena.numel() and start < edge < start + nbytes
        return arena, arena[off:off + nbytes].view(dt).view(shape)
    keep, tens = zip(*[straddling((B, H, N, d), dtype) for _ in range(4)])
    Q, K, V, O = tens
    _, L = straddling((B, H, N, 1), dtype)
    for t, s_ in zip((Q, K, V), src):
        t.copy_(s_)
    _lib.fa2_fwd(Q, K, V, O, L, _lib.FA2_DTYPE_BF16, causal=True, variant=_lib.VARIANTS[variant])
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(*(t.float() for t in (Q, K, V)), scale=1.0, is_causal=True)
    assert (O.float() - ref).abs().max().item() <= O_TOL[dtype]
    S = (Q.float() @ K.float().transpose(-1, -2)).masked_fill(~torch.ones(N, N, dtype=torch.bool, device=DEV).tril(), float("-inf"))
    L_ref = torch.logsumexp(S, dim=-1, keepdim=True) * math.log2(math.e)
    assert (L.float() - L_ref).abs().max().item() <= 1.01 * ulp(dtype, L_ref.abs().max().item())
    del keep


@pytest.mark.parametrize("variant", ["a64", "a16", "mfma16h", "mfma16d_w4"])
def test_nan_and_inf_inputs_propagate_like_the_oracle(oracle, variant):
    """A NaN in Q poisons its row, a NaN in V its column of the rows that see it, a NaN key every row that sees it; +Inf in V
    gives +Inf / NaN as the oracle says.  (The default kernels are built with -fno-honor-nans: this pins the behaviour.)"""
    dtype = torch.bfloat16
    Q, K, V = rand3((1, 1, 256, 128), dtype, seed=3, spread=0.5)
    Q[0, 0, 10, 3] = float("nan")
    V[0, 0, 50, 7] = float("nan")
    V[0, 0, 60, 9] = float("inf")
    O_ref, L_ref = oracle_fwd(oracle, Q, K, V, dtype, False)
    O, L = a64(Q, K, V, False, variant=variant)
    nan_ref, nan = torch.isnan(O_ref), torch.isnan(O.float())
    if variant == "a64":
        # documented deviation (include/fa2_fwd.h): a64 sums the rows of P on the matrix pipe, where the P of query q + 16 (or
        # q - 16, the other query of the lane pair in q's 32-row block) meets a zero weight -- 0 * NaN = NaN: a NaN QUERY row
        # also turns that one partner row into NaN.  Nothing else changes (NaN / Inf in K and V propagate as in the oracle).
        partner = nan_ref.clone()
        partner[0, 0, 10 ^ 16] = True
        assert torch.equal(nan, partner), (nan.sum().item(), partner.sum().item())
        nan_ref = partner
        L_ref = L_ref.clone()
        L_ref.view(-1)[10 ^ 16] = float("nan")
    assert torch.equal(nan, nan_ref), (variant, nan.sum().item(), nan_ref.sum().item())
    inf_ref = torch.isinf(O_ref) & ~nan_ref
    assert torch.equal(torch.isinf(O.float()), inf_ref)
    ok = ~(nan_ref | inf_ref)
    assert (O.float()[ok] - O_ref[ok]).abs().max() <= O_TOL[dtype]
    assert torch.equal(torch.isnan(L.float().flatten()), torch.isnan(L_ref.flatten()))
    # causal (our extension): the oracle skips masked keys, SDPA multiplies their V by a zero weight (0 * NaN = NaN).  The
    # kernels skip whole tiles above the diagonal and multiply by zero inside the diagonal tile: between the two.
    O_ref, _ = oracle_fwd(oracle, Q, K, V, dtype, True)
    O, _ = a64(Q, K, V, True, variant=variant)
    sdpa = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0, is_causal=True)
    bad, lo, hi = ~torch.isfinite(O.float()), ~torch.isfinite(O_ref), ~torch.isfinite(sdpa)
    if variant == "a64":
        hi[0, 0, 10 ^ 16] = True                              # the partner row of the NaN query, as above
    assert (bad | ~lo).all() and (hi | ~bad).all()            # lo subset of bad subset of hi
    assert (O.float()[~bad] - O_ref[~bad]).abs().max() <= O_TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@both
def test_ragged_N_vs_oracle(oracle, dtype, causal, variant):
    """N above 256 that is not a multiple of 256: the "ragged" kernels (range-checked descriptors with every offset in the VGPR
    operand: rows past N load as zeros and are never stored; non-causal: the key tail of the job's last four tiles is masked --
    a partly real tile exactly, wholly unreal ones through +inf as running maximum).  Element-wise against the oracle's
    deferred-maximum mode; O and L sit in canary arenas."""
    for N in (257, 300, 448, 600, 1000, 1333, 2049):
        B, H = 2, 3
        Q, K, V = rand3((B, H, N, 128), dtype, seed=N)
        arena_o = torch.full((B, H, N + 8, 128), 768.0, dtype=dtype, device=DEV)
        arena_l = torch.full((B, H, N + 8, 1), 768.0, dtype=dtype, device=DEV)
        O, L = arena_o[:, :, :N], arena_l[:, :, :N]
        _lib.fa2_fwd(Q.to(DEV), K.to(DEV), V.to(DEV), O, L, fa.convert_triton_dtype(dtype), causal=causal, variant=_lib.VARIANTS[variant])
        torch.cuda.synchronize()
        assert (arena_o[:, :, N:].float() == 768.0).all() and (arena_l[:, :, N:].float() == 768.0).all(), N
        f = lambda t: t.float().numpy()
        O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, G=32, B_c=64,
                                               thr=60.0 if dtype == torch.bfloat16 else 15.875, sum_rounded=True)
        O_ref, L_ref = torch.from_numpy(O_ref), torch.from_numpy(L_ref)
        assert (O.cpu().float() == O_ref).float().mean() >= 0.99, N
        check(O.cpu(), L.cpu(), O_ref, L_ref, dtype)


def test_two_streams_at_once():
    """the launcher keeps no per-launch state: two a64 problems in flight on two streams give the bits of the serial runs"""
    torch.manual_seed(3)
    a = [torch.randn(2, 16, 2048, 128, device=DEV).bfloat16() for _ in range(3)]     # causal, 256 jobs
    b = [torch.randn(1, 24, 1000, 128, device=DEV).half() for _ in range(3)]         # ragged, non-causal
    Oa, La = fa.flash_attention_forward(*a, DEV, causal=True, variant="a64")
    Ob, Lb = fa.flash_attention_forward(*b, DEV, causal=False, variant="a64")
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(s1):
            Oa2, La2 = fa.flash_attention_forward(*a, DEV, causal=True, variant="a64")
        with torch.cuda.stream(s2):
            Ob2, Lb2 = fa.flash_attention_forward(*b, DEV, causal=False, variant="a64")
    torch.cuda.synchronize()
    assert torch.equal(Oa, Oa2) and torch.equal(La, La2) and torch.equal(Ob, Ob2) and torch.equal(Lb, Lb2)


# ----------------------------------------------------------------------------- head size 64: the generated kernel a64d
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
def test_a64d_head_size_64(oracle, dtype, causal):
    """the generated kernel at d = 64 (asm/fa2_a64d_gen.py): seeded inputs against the oracle, many jobs per workgroup against SDPA,
    (B, N, H, d)-strided inputs, element-wise against the oracle's deferred-maximum mode (>= 99 % of O bit-identical), and
    bit-identical head shards (causal: light jobs walk downwards as a function of the query block alone)"""
    for shape in ((1, 1, 256), (2, 3, 512), (1, 2, 768), (1, 5, 1024)):
        B, H, N = shape
        Q, K, V = rand3((B, H, N, 64), dtype, seed=N + H)
        f = lambda t: t.float().contiguous().numpy()
        O_ref, L_ref = oracle.forward(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, B_r=64, B_c=64)
        check(*a64(Q, K, V, causal, variant="a64d"), torch.from_numpy(O_ref), torch.from_numpy(L_ref), dtype)
    Q, K, V = rand3((3, 48, 1536, 64), dtype, seed=9)        # 864 jobs on 256 workgroups, B * H not a power of two
    O, L = a64(Q, K, V, causal, variant="a64d")
    ref = torch.nn.functional.scaled_dot_product_attention(Q.to(DEV).float(), K.to(DEV).float(), V.to(DEV).float(), scale=1.0, is_causal=causal).cpu()
    assert torch.isfinite(O.float()).all() and (O.float() - ref).abs().max() <= O_TOL[dtype]
    parts = [a64(Q[:, h0:h0 + 12].contiguous(), K[:, h0:h0 + 12].contiguous(), V[:, h0:h0 + 12].contiguous(), causal, variant="a64d")[0]
             for h0 in range(0, 48, 12)]
    assert torch.equal(torch.cat(parts, dim=1), O)
    gen = torch.Generator().manual_seed(21)
    Qs, Ks, Vs = (torch.randn(2, 512, 3, 64, generator=gen).to(dtype).transpose(1, 2) for _ in range(3))
    Os, _ = a64(Qs, Ks, Vs, causal, variant="a64d")
    Oc, _ = a64(Qs.contiguous(), Ks.contiguous(), Vs.contiguous(), causal, variant="a64d")
    assert torch.equal(Os, Oc)
    Q, K, V = rand3((1, 2, 512, 64), dtype, seed=77)
    K[:, :, 300] = (Q[:, :, 200].float() * 0.9).to(dtype)
    O, L = a64(Q, K, V, causal, variant="a64d")
    f = lambda t: t.float().numpy()
    O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), ORACLE_NAME[dtype], causal=causal, G=32, B_c=64,
                                           thr=60.0 if dtype == torch.bfloat16 else 15.875, sum_rounded=True)
    assert (O.float() == torch.from_numpy(O_ref)).float().mean().item() >= 0.99
    assert _lib.query_tile(4096, 64, _lib.FA2_DTYPE_BF16, causal, B=8, H=16)[0] == _lib.VARIANT_A64D
    assert _lib.query_tile(4000, 64, _lib.FA2_DTYPE_BF16, causal, B=8, H=16)[0] == _lib.VARIANT_A64D      # (its ragged form)
    for N in (300, 1000, 2049):     # ragged N: canaries behind the tensors, the oracle's deferred mode element by element
        Qr, Kr, Vr = rand3((2, 3, N, 64), dtype, seed=N)
        arena_o = torch.full((2, 3, N + 8, 64), 768.0, dtype=dtype, device=DEV)
        arena_l = torch.full((2, 3, N + 8, 1), 768.0, dtype=dtype, device=DEV)
        Or_, Lr_ = arena_o[:, :, :N], arena_l[:, :, :N]
        _lib.fa2_fwd(Qr.to(DEV), Kr.to(DEV), Vr.to(DEV), Or_, Lr_, fa.convert_triton_dtype(dtype), causal=causal, variant=_lib.VARIANT_A64D)
        torch.cuda.synchronize()
        assert (arena_o[:, :, N:].float() == 768.0).all() and (arena_l[:, :, N:].float() == 768.0).all(), N
        O_ref, L_ref = oracle.forward_deferred(f(Qr), f(Kr), f(Vr), ORACLE_NAME[dtype], causal=causal, G=32, B_c=64,
                                               thr=60.0 if dtype == torch.bfloat16 else 15.875, sum_rounded=True)
        assert (Or_.cpu().float() == torch.from_numpy(O_ref)).float().mean().item() >= 0.99, N
        check(Or_.cpu(), Lr_.cpu(), torch.from_numpy(O_ref), torch.from_numpy(L_ref), dtype)


def test_first_launch_of_the_process_under_stream_capture():
    """"safe under stream capture" -- for the default kernel too, whose first launch on a device loads its code
    object (hipModuleLoadData): a fresh process captures variant "a64" into a HIP graph with NO warm-up call, replays it on new
    contents of the same buffers and compares with an eager launch, bit for bit."""
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import torch, flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(11)
mk = lambda: torch.randn(2, 4, 512, 128, generator=gen).to(torch.bfloat16).to(dev)
Q, K, V = mk(), mk(), mk()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    O_g, L_g = fa.flash_attention_forward(Q, K, V, dev, causal=True, variant="a64")
for t in (Q, K, V):
    t.copy_(mk())
g.replay()
torch.cuda.synchronize()
O_e, L_e = fa.flash_attention_forward(Q, K, V, dev, causal=True, variant="a64")
torch.cuda.synchronize()
assert torch.isfinite(O_e.float()).all() and O_e.float().abs().max() > 0.5
assert torch.equal(O_g, O_e) and torch.equal(L_g, L_e)
print("CAPTURE_OK")
"""
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CAPTURE_OK" in out.stdout, out.stderr[-2000:]


def test_unsupported_shapes_raise_and_auto_falls_back():
    Q, K, V = rand3((1, 2, 200, 128), torch.bfloat16, seed=1)
    with pytest.raises(TypeError):
        a64(Q, K, V)
    O, _ = a64(Q, K, V, variant="auto")
    ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0)
    assert (O.float() - ref).abs().max() <= O_TOL[torch.bfloat16]


# ----------------------------------------------------------------------------- full per-GPU shards of BASELINE.json configs[3], [4]
def test_c4_per_gpu_shard_properties():
    """configs[3]: B8 H32 N8192 d128 bf16 head-sharded over 4 GPUs -> one GPU's shard B8 H8 N8192, non-causal"""
    dtype, B, H, N = torch.bfloat16, 8, 8, 8192
    torch.manual_seed(42)
    Q, K, V = (torch.randn(B, H, N, 128, device=DEV).to(dtype) for _ in range(3))
    assert _lib.query_tile(N, 128, _lib.FA2_DTYPE_BF16, False)[0] == _lib.VARIANT_A16    # (the generated kernel in its 16x16x32 form)
    O, L = fa.flash_attention_forward(Q, K, V, DEV)
    O2, L2 = fa.flash_attention_forward(Q, K, V, DEV)
    assert torch.equal(O, O2) and torch.equal(L, L2)                                   # run-to-run bit equality
    parts = [fa.flash_attention_forward(Q[:, h0:h0 + 2].contiguous(), K[:, h0:h0 + 2].contiguous(),
                                        V[:, h0:h0 + 2].contiguous(), DEV)[0] for h0 in range(0, H, 2)]
    assert torch.equal(torch.cat(parts, dim=1), O)                                     # bit-identical head indexing
    O1, _ = fa.flash_attention_forward(Q, K, torch.ones_like(V), DEV)
    assert (O1.float() - 1).abs().max().item() <= O_TOL[dtype]                         # V = 1 => O = 1
    for (b, h) in ((0, 0), (7, 7), (3, 5)):                                            # fp32 attention of the bf16 inputs
        S = Q[b, h].float() @ K[b, h].float().T
        ref = torch.softmax(S, dim=-1) @ V[b, h].float()
        assert (O[b, h].float() - ref).abs().max().item() <= O_TOL[dtype]
        lse2 = torch.logsumexp(S, dim=-1) * math.log2(math.e)
        assert (L[b, h].float().flatten() - lse2).abs().max().item() <= 1.01 * ulp(dtype, lse2.abs().max().item())


def test_c5_full_per_gpu_shard_properties():
    """configs[4]: B16 H64 N16384 d128 fp8 over 8 GPUs -> one GPU's FULL shard B16 H8 N16384 e4m3 (1 GiB of Q, K, V, O)"""
    dtype, B, H, N = torch.float8_e4m3fn, 16, 8, 16384
    torch.manual_seed(42)
    Q, K, V = ((torch.randn(B, H, N, 128, device=DEV) * 0.5).to(dtype) for _ in range(3))
    O, L = fa.flash_attention_forward(Q, K, V, DEV)
    O2, L2 = fa.flash_attention_forward(Q, K, V, DEV)
    u8 = lambda t: t.view(torch.uint8)
    assert torch.equal(u8(O), u8(O2)) and torch.equal(u8(L), u8(L2))
    parts = [fa.flash_attention_forward(Q[b0:b0 + 4].contiguous(), K[b0:b0 + 4].contiguous(), V[b0:b0 + 4].contiguous(), DEV)[0]
             for b0 in range(0, B, 4)]
    assert torch.equal(u8(torch.cat(parts, dim=0)), u8(O))                             # batch shards == the full run
    O1, _ = fa.flash_attention_forward(Q, K, torch.ones(B, H, N, 128, device=DEV).to(dtype), DEV)
    assert (O1.float() == 1).all()
    assert torch.isfinite(O.float()).all() and torch.isfinite(L.float()).all()
    for (b, h) in ((0, 0), (15, 7)):
        S = Q[b, h].double() @ K[b, h].double().T
        ref = (torch.softmax(S, dim=-1) @ V[b, h].double()).float()
        rel = ((O[b, h].float() - ref).abs() / ref.abs().clamp(min=0.05)).flatten()
        assert rel.median() <= 2.0 ** -4 and rel.kthvalue(int(0.99 * rel.numel())).values <= 3 * 2.0 ** -3
        lse2 = (torch.logsumexp(S, dim=-1) * math.log2(math.e)).float()
        assert ((L[b, h].float().flatten() - lse2).abs() <= 2.0 ** -3 * lse2.abs() + 1e-3).all()
