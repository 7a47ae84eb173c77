"""GPU parity of the native backward (include/fa2_bwd.h) -- SURVEY.md section 8 row f1.

Everything goes through the C ABI (flash_attention_backward -> _lib.fa2_bwd -> libfa2_hip.so).  Checked against
  (a) tests/golden/bwd_*.npz: outputs of the reference's own bwd_D_kernel + bwd_kernel (Triton interpreter) and of
      autograd through fp64 SDPA(scale=1);
  (b) the CPU oracle (oracle/fa2_oracle_bwd.c, pinned to (a) by tests/test_oracle_bwd.py) and an fp64 numpy
      restatement, on seeded inputs over ragged shapes, both dtypes families, causal and scale;
  (c) live torch autograd on the same device at larger sizes, with the reference's own tolerances
      (src/test_correctness.py:60-62: atol 9e-4 / 7e-4 / 7e-5, rtol 1e-5 for fp32), determinism, linearity in dO.
Tolerances.  fp32: the reference's.  f16 / bf16: the gradients are compared with the fp64 truth of the ROUNDED inputs,
within `REL[dtype] * max|truth|` (one output rounding = 2^-11 / 2^-8 relative plus the P / dS roundings before the
second contractions); the reference's own 16-bit scheme (running sums rounded to the I/O dtype at every block, L read
back in the I/O dtype) is less accurate than that -- the oracle comparison uses `REL_ORACLE`.
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

import flash_attention_dlrs_amd as fa  # noqa: E402
from flash_attention_dlrs_amd import flash_attention_torch as ft  # noqa: E402

DEV = torch.device("cuda:0")
ORACLE_NAME = {torch.float32: "float32", torch.float16: "float16", torch.bfloat16: "bfloat16"}
REL = {torch.float16: 4e-3, torch.bfloat16: 2.5e-2}
REL_ORACLE = {torch.float16: 2e-2, torch.bfloat16: 1.2e-1}


def variants_for(dtype, d):
    v = ["auto", "generic"]
    if dtype in (torch.float16, torch.bfloat16) and d in (64, 128):
        v.append("mfma16")
    if dtype == torch.float32 and d in (64, 128):
        v.append("mfma32")
    return v


def hip_fwd_bwd(Q, K, V, dO, causal=False, scale=1.0, variant="auto"):
    Qd, Kd, Vd, dOd = (t.to(DEV) for t in (Q, K, V, dO))
    O, L = fa.flash_attention_forward(Qd, Kd, Vd, DEV, causal=causal, scale=scale)
    dQ, dK, dV = fa.flash_attention_backward(Qd, Kd, Vd, O, dOd, L, DEV, causal=causal, scale=scale, variant=variant)
    torch.cuda.synchronize()
    return tuple(t.cpu() for t in (dQ, dK, dV)), O.cpu(), L.cpu()


def bf16(u16):
    return torch.from_numpy(u16.view(np.int16).copy()).view(torch.bfloat16)


# ----------------------------------------------------------------------------- (a) golden vectors
@pytest.mark.parametrize("name", ["bwd_test_torch_f32_seed5", "bwd_c1_f32_seed11"])
def test_golden_fp32(name):
    g = load_golden(name)
    Q, K, V, dO = (torch.from_numpy(g[k]) for k in ("Q", "K", "V", "dO"))
    for variant in variants_for(torch.float32, Q.shape[-1]):
        (dQ, dK, dV), _, _ = hip_fwd_bwd(Q, K, V, dO, variant=variant)
        for k, a, atol in (("dQ", dQ, 9e-4), ("dK", dK, 7e-4), ("dV", dV, 7e-5)):
            assert torch.allclose(torch.from_numpy(g[f"{k}_sdpa"]), a, atol=atol, rtol=1e-5), (variant, k)   # reference's bar
            assert (a - torch.from_numpy(g[f"{k}_ref"])).abs().max() < 4e-4, (variant, k)                      # vs the reference kernels


def test_golden_fp16():
    g = load_golden("bwd_c1_f16_seed12")
    Q, K, V, dO = (torch.from_numpy(g[k]) for k in ("Q", "K", "V", "dO"))
    for variant in variants_for(torch.float16, 64):
        (dQ, dK, dV), _, _ = hip_fwd_bwd(Q, K, V, dO, variant=variant)
        for k, a in (("dQ", dQ), ("dK", dK), ("dV", dV)):
            truth = torch.from_numpy(g[f"{k}_sdpa"])
            assert a.dtype == torch.float16
            assert (a.float() - truth).abs().max() <= REL[torch.float16] * truth.abs().max(), (variant, k)
            # the reference kernel's fp16 output sits further from the truth than ours; both within REL_ORACLE
            ref = torch.from_numpy(g[f"{k}_ref"]).float()
            assert (a.float() - ref).abs().max() <= REL_ORACLE[torch.float16] * truth.abs().max(), (variant, k)


@pytest.mark.parametrize("causal", [False, True])
def test_golden_bf16_and_causal(causal):
    sfx = "_causal" if causal else ""
    g = load_golden("bwd_c1_bf16_seed14")
    Q, K, V, dO = (bf16(g[k]) for k in ("Q", "K", "V", "dO"))
    for variant in variants_for(torch.bfloat16, 64):
        (dQ, dK, dV), _, _ = hip_fwd_bwd(Q, K, V, dO, causal=causal, variant=variant)
        for k, a in (("dQ", dQ), ("dK", dK), ("dV", dV)):
            truth = torch.from_numpy(g[f"{k}_sdpa{sfx}"])
            assert (a.float() - truth).abs().max() <= REL[torch.bfloat16] * truth.abs().max(), (variant, k)
    g = load_golden("bwd_c1_f32_causal_seed13")
    Q, K, V, dO = (torch.from_numpy(g[k]) for k in ("Q", "K", "V", "dO"))
    (dQ, dK, dV), _, _ = hip_fwd_bwd(Q, K, V, dO, causal=causal)
    for k, a, atol in (("dQ", dQ, 9e-4), ("dK", dK, 7e-4), ("dV", dV, 7e-5)):
        assert torch.allclose(torch.from_numpy(g[f"{k}_sdpa{sfx}"]), a, atol=atol, rtol=1e-5), k


# ----------------------------------------------------------------------------- (b) seeded vs oracle
SHAPES = [(1, 1, 16, 16), (2, 3, 48, 32), (1, 2, 128, 64), (2, 2, 256, 128), (1, 2, 1, 64), (1, 1, 17, 64),
          (2, 1, 100, 128), (1, 2, 130, 64), (1, 1, 321, 128), (1, 2, 200, 32), (1, 8, 384, 128), (1, 1, 64, 256)]


def _rand4(shape, dtype, seed, spread=1.0):
    gen = torch.Generator().manual_seed(seed)
    return tuple((torch.randn(*shape, generator=gen) * spread).to(dtype) for _ in range(4))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_seeded_vs_oracle(oracle, dtype, causal, shape):
    spread = 1.0 if dtype == torch.float32 else 0.6
    Q, K, V, dO = _rand4(shape, dtype, seed=sum(shape) * 7 + int(causal), spread=spread)
    f = lambda t: t.float().numpy()
    truth = oracle.grads_f64(f(Q), f(K), f(V), f(dO), causal=causal)
    N = shape[2]
    tile = next(t for t in (32, 16, 8, 4, 2, 1) if N % t == 0)
    for variant in variants_for(dtype, shape[-1]):
        (dQ, dK, dV), O, L = hip_fwd_bwd(Q, K, V, dO, causal=causal, variant=variant)
        for k, a, t in (("dQ", dQ, truth[0]), ("dK", dK, truth[1]), ("dV", dV, truth[2])):
            assert a.shape == Q.shape and a.dtype == dtype
            err = np.abs(a.double().numpy() - t).max()
            bound = (2e-4 if dtype == torch.float32 else REL[dtype]) * max(1.0, np.abs(t).max())
            assert err <= bound, (variant, k, err, bound)
        if variant == "auto":  # the C restatement of the reference kernels, fed with OUR forward's O and L
            o = oracle.backward(f(Q), f(K), f(V), f(O), f(dO), f(L), ORACLE_NAME[dtype], causal=causal, B_r=tile, B_c=tile)
            for k, a, r, t in (("dQ", dQ, o[0], truth[0]), ("dK", dK, o[1], truth[1]), ("dV", dV, o[2], truth[2])):
                bound = (4e-4 if dtype == torch.float32 else REL_ORACLE[dtype]) * max(1.0, np.abs(t).max())
                assert np.abs(a.float().numpy() - r).max() <= bound, (k, "oracle")


@pytest.mark.parametrize("dtype,variant", [(torch.float32, "generic"), (torch.bfloat16, "mfma16"), (torch.float16, "generic")])
def test_scale_extension(oracle, dtype, variant):
    Q, K, V, dO = _rand4((1, 2, 192, 128), dtype, seed=21)
    scale = 1.0 / math.sqrt(128)
    f = lambda t: t.float().numpy()
    truth = oracle.grads_f64(f(Q), f(K), f(V), f(dO), causal=True, scale=scale)
    (dQ, dK, dV), _, _ = hip_fwd_bwd(Q, K, V, dO, causal=True, scale=scale, variant=variant)
    for a, t in zip((dQ, dK, dV), truth):
        bound = (2e-5 if dtype == torch.float32 else REL[dtype]) * max(1.0, np.abs(t).max())
        assert np.abs(a.double().numpy() - t).max() <= bound


def test_fp64_generic(oracle):
    Q, K, V, dO = _rand4((1, 2, 72, 32), torch.float64, seed=4)
    truth = oracle.grads_f64(Q.numpy(), K.numpy(), V.numpy(), dO.numpy(), causal=True)
    (dQ, dK, dV), _, _ = hip_fwd_bwd(Q, K, V, dO, causal=True)
    for a, t in zip((dQ, dK, dV), truth):
        assert np.abs(a.numpy() - t).max() < 1e-11


def test_strided_inputs_and_padding():
    """(B, N, H, d) storage viewed as (B, H, N, d) (the gradients inherit Q's strides, torch.py:101-103) and a head
    size that the glue pads (d = 40 -> 64, torch.py:95-99)."""
    torch.manual_seed(3)
    Qs, Ks, Vs, Gs = (torch.randn(2, 96, 3, 40, device=DEV, requires_grad=True) for _ in range(4))
    Q, K, V = (t.transpose(1, 2) for t in (Qs, Ks, Vs))
    dO = Gs.detach().transpose(1, 2)
    O = fa.FlashAttention.apply(Q, K, V)
    g = torch.autograd.grad(O, (Qs, Ks, Vs), dO)
    O_t = torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1)
    g_t = torch.autograd.grad(O_t, (Qs, Ks, Vs), dO)
    for a, b, atol in zip(g, g_t, (9e-4, 7e-4, 7e-5)):
        assert a.shape == b.shape and torch.allclose(b, a, atol=atol, rtol=1e-5)


def test_fp8_backward_is_rejected():
    x = torch.randn(1, 1, 64, 128, device=DEV).to(torch.float8_e5m2)
    O, L = fa.flash_attention_forward(x, x, x, DEV)
    with pytest.raises(TypeError):
        fa.flash_attention_backward(x, x, x, O, x, L, DEV)


# ----------------------------------------------------------------------------- reference harness counterparts
def test_reference_gradcheck_script():
    """Counterpart of src/test_torch.py: gradcheck of both autograd classes at B2 H2 N32 d128 fp32, seed 5, with the
    reference's settings (eps=2e-2, atol=1e-2, rtol=1e-2, nondet_tol=1e-4)."""
    torch.manual_seed(5)
    Q = torch.randn(2, 2, 32, 128, dtype=torch.float32, device=DEV, requires_grad=True)
    K = torch.randn_like(Q, requires_grad=True)
    V = torch.randn_like(Q, requires_grad=True)
    for cls in (fa.FlashAttention, fa.FlashAttentionDeterministic):
        assert torch.autograd.gradcheck(cls.apply, (Q, K, V), eps=2e-2, atol=1e-2, rtol=1e-2, nondet_tol=1e-4)


# ----------------------------------------------------------------------------- (c) larger sizes, live autograd
def _autograd_ref(Q, K, V, dO, causal):
    q, k, v = (t.detach().float().requires_grad_(True) for t in (Q, K, V))
    with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.MATH):
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0, is_causal=causal)
    return torch.autograd.grad(o, (q, k, v), dO.float())


@pytest.mark.parametrize("shape,dtype,causal", [((2, 8, 1024, 64), torch.float16, False),     # BASELINE.json configs[1]
                                                ((1, 8, 2048, 128), torch.bfloat16, True),    # configs[2] shape, fewer heads
                                                ((1, 4, 1000, 128), torch.bfloat16, False),   # ragged
                                                ((4, 4, 256, 128), torch.float32, False)])    # reference test shape (test_correctness.py)
def test_larger_sizes_vs_live_autograd_and_properties(shape, dtype, causal):
    torch.manual_seed(11)
    Q, K, V, dO = ((torch.randn(*shape, device=DEV) * (1.0 if dtype == torch.float32 else 0.5)).to(dtype) for _ in range(4))
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
    g = fa.flash_attention_backward(Q, K, V, O, dO, L, DEV, causal=causal)
    ref = _autograd_ref(Q, K, V, dO, causal)
    for k, a, r, atol in zip("QKV", g, ref, (9e-4, 7e-4, 7e-5)):
        if dtype == torch.float32:
            assert torch.allclose(r, a, atol=atol, rtol=1e-5), k
        else:
            assert (a.float() - r).abs().max() <= REL[dtype] * r.abs().max(), k
    # deterministic: a second run is bit-identical (no cross-workgroup sums)
    g2 = fa.flash_attention_backward(Q, K, V, O, dO, L, DEV, deterministic=True, causal=causal)
    for a, b2 in zip(g, g2):
        assert torch.equal(a, b2)
    # linear in dO: grads(2 dO) == 2 grads(dO) exactly -- power-of-two scaling commutes with every rounding, except
    # in fp16 where small dS values are subnormal (below 2^-14) and gain a bit when doubled
    g3 = fa.flash_attention_backward(Q, K, V, O, dO * 2, L, DEV, causal=causal)
    for a, b3 in zip(g, g3):
        if dtype == torch.float16:
            assert (a.float() * 2 - b3.float()).abs().max() <= 2e-3 * b3.float().abs().max()
        else:
            assert torch.equal(a * 2, b3)
    # dO = 0 -> all gradients 0; rows of dQ for which dO is a multiple of ... (skip)
    g0 = fa.flash_attention_backward(Q, K, V, O, torch.zeros_like(dO), L, DEV, causal=causal)
    for a in g0:
        assert (a == 0).all()
