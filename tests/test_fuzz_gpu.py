"""Randomised shape sweep of the forward and the backward on the GPU against fp64 truth (torch autograd through a plain
fp64 softmax(QK^T)V on the same device): catches indexing mistakes that fixed shape lists miss -- N around every tile
boundary (32 / 64 / 128 / 256 and the causal tile pairs), B*H multiples of 8 and not (XCD mapping), d in {16..256},
both 16-bit dtypes and fp32, causal and not, every kernel variant the problem supports.  Seeded: the same cases every run."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

import flash_attention_dlrs_amd as fa  # noqa: E402

DEV = torch.device("cuda:0")
REL = {torch.float32: 3e-5, torch.float16: 4e-3, torch.bfloat16: 2.5e-2}


def truth(Q, K, V, dO, causal):
    q, k, v = (t.double().detach().requires_grad_(True) for t in (Q, K, V))
    S = q @ k.transpose(-1, -2)
    if causal:
        N = Q.shape[2]
        S = S.masked_fill(~torch.ones(N, N, dtype=torch.bool, device=Q.device).tril(), float("-inf"))
    O = torch.softmax(S, dim=-1) @ v
    g = torch.autograd.grad(O, (q, k, v), dO.double())
    return O.detach(), g


def cases():
    rng = random.Random(20261004)
    edges = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 640, 769, 1023, 1025]
    out = []
    for k in range(48):
        N = rng.choice(edges) if k % 3 else rng.randint(1, 1100)
        d = rng.choice([16, 32, 64, 64, 128, 128, 128, 256])
        B, H = rng.choice([(1, 1), (1, 3), (2, 4), (1, 8), (3, 8), (2, 5)])
        dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
        out.append((B, H, N, d, dtype, bool(k & 1)))
    return out


@pytest.mark.parametrize("B,H,N,d,dtype,causal", cases(), ids=lambda v: str(v).replace("torch.", ""))
def test_random_shape(B, H, N, d, dtype, causal):
    g = torch.Generator().manual_seed(B * 1000003 + H * 10007 + N * 101 + d)
    spread = 1.0 if dtype == torch.float32 else 0.6
    Q, K, V, dO = ((torch.randn(B, H, N, d, generator=g) * spread).to(dtype).to(DEV) for _ in range(4))
    O_t, g_t = truth(Q, K, V, dO, causal)
    fwd_variants = ["auto", "generic"]
    if dtype != torch.float32 and d in (64, 128):
        fwd_variants += ["mfma16d", "mfma16d_w4", "mfma16h", "mfma16h_w4", "mfma16k", "mfma16k_r2k2"] + ([] if d == 128 else ["mfma16k_r2k4"])
    if dtype != torch.float32 and d == 128 and N >= 256:
        fwd_variants += ["a64", "a16"]      # (their plain forms when N is a multiple of 256, the ragged ones otherwise)
    if dtype != torch.float32 and d == 64 and N >= 256:
        fwd_variants += ["a64d"]     # (the generated kernel at head size 64: plain or ragged form)
    if dtype == torch.float32 and d in (64, 128):
        fwd_variants += ["mfma32"]
    for v in fwd_variants:
        O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal, variant=v)
        err = (O.double() - O_t).abs().max().item()
        assert err <= REL[dtype] * max(1.0, O_t.abs().max().item()) * (4 if dtype == torch.float32 else 1), ("fwd", v, err)
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
    bwd_variants = ["auto", "generic"]
    for v in bwd_variants:
        grads = fa.flash_attention_backward(Q, K, V, O, dO, L, DEV, causal=causal, variant=v)
        for name, a, t in zip("QKV", grads, g_t):
            err = (a.double() - t).abs().max().item()
            bound = (2e-4 if dtype == torch.float32 else REL[dtype]) * max(1.0, t.abs().max().item())
            assert err <= bound, ("bwd", v, name, err, bound)
    assert math.isfinite(L.float().abs().max().item())


def a64_cases():
    rng = random.Random(4242)
    out = []
    for k in range(24):
        N = rng.choice([256, 257, 300, 511, 512, 513, 700, 767, 768, 769, 1000, 1024, 1025, 1279, 1500, 2047, 2048, 2049, 2500]) if k % 2 else rng.randint(256, 3000)
        B, H = rng.choice([(1, 1), (1, 3), (2, 4), (1, 8), (3, 8), (2, 5), (4, 16), (1, 40)])
        out.append((B, H, N, rng.choice([torch.bfloat16, torch.float16]), bool(k & 1), rng.choice([1.0, 1.0, 128 ** -0.5, 0.3]),
                    rng.choice(["contiguous", "bnhd", "padded_rows"])))
    return out


@pytest.mark.parametrize("variant", ["a64", "a16"])
@pytest.mark.parametrize("B,H,N,dtype,causal,scale,layout", a64_cases(), ids=lambda v: str(v).replace("torch.", ""))
def test_random_a64_problem(B, H, N, dtype, causal, scale, layout, variant):
    """the generated assembly kernels (both matrix shapes) over random (B, H, N) -- one job to many per workgroup, N ragged or not -- scales and
    storage layouts ((B, N, H, d) permuted views, rows padded to 136 elements), against fp64 attention on the device"""
    g = torch.Generator().manual_seed(N * 7919 + B * 31 + H)
    mk = {"contiguous": lambda: (torch.randn(B, H, N, 128, generator=g) * 0.7).to(dtype).to(DEV),
          "bnhd": lambda: (torch.randn(B, N, H, 128, generator=g) * 0.7).to(dtype).to(DEV).transpose(1, 2),
          "padded_rows": lambda: (torch.randn(B, H, N, 136, generator=g) * 0.7).to(dtype).to(DEV)[..., :128]}[layout]
    Q, K, V = mk(), mk(), mk()
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal, scale=scale, variant=variant)
    q, k, v = (t.double() for t in (Q, K, V))
    S = (q @ k.transpose(-1, -2)) * scale
    if causal:
        S = S.masked_fill(~torch.ones(N, N, dtype=torch.bool, device=DEV).tril(), float("-inf"))
    O_t = torch.softmax(S, dim=-1) @ v
    L_t = torch.logsumexp(S, dim=-1, keepdim=True) * math.log2(math.e)
    assert (O.double() - O_t).abs().max().item() <= REL[dtype] * max(1.0, O_t.abs().max().item())
    lu = 2.0 ** (math.floor(math.log2(max(L_t.abs().max().item(), 1e-9))) - (7 if dtype == torch.bfloat16 else 10))
    assert (L.double() - L_t).abs().max().item() <= 1.01 * lu


def a8_cases():
    rng = random.Random(8008)
    out = []
    for k in range(16):
        N = 256 * rng.choice([1, 2, 3, 4, 5, 8, 9, 12]) + (rng.choice([0, 1, 44, 100, 255]) if k % 3 == 2 else 0)
        B, H = rng.choice([(1, 1), (1, 3), (2, 4), (1, 8), (3, 8), (2, 5), (4, 16), (1, 40)])
        out.append((B, H, N, rng.choice([torch.float8_e4m3fn, torch.float8_e5m2]), bool(k & 1), rng.choice([1.0, 1.0, 128 ** -0.5, 0.3]),
                    rng.choice(["contiguous", "bnhd", "padded_rows"]), rng.choice([0.4, 0.7, 1.0])))
    return out


@pytest.mark.parametrize("B,H,N,dtype,causal,scale,layout,spread", a8_cases(), ids=lambda v: str(v).replace("torch.", ""))
def test_random_a8_problem(B, H, N, dtype, causal, scale, layout, spread):
    """the generated fp8 kernel (block-scaled P.V, integer running maximum) over random (B, H, N), causal and not, scales, input
    spreads (the maximum moves rarely ... in most steps) and storage layouts, against fp64 attention of the fp8 inputs on the device:
    the statistical bars of the fp8 tests (median relative error half an ulp, 99th percentile three, L within an fp8 step)"""
    g = torch.Generator().manual_seed(N * 7919 + B * 31 + H)
    mk = {"contiguous": lambda: (torch.randn(B, H, N, 128, generator=g) * spread).to(dtype).to(DEV),
          "bnhd": lambda: (torch.randn(B, N, H, 128, generator=g) * spread).to(dtype).to(DEV).transpose(1, 2),
          "padded_rows": lambda: (torch.randn(B, H, N, 144, generator=g) * spread).to(dtype).to(DEV)[..., :128]}[layout]
    Q, K, V = mk(), mk(), mk()
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal, scale=scale, variant="a8")
    q, k, v = (t.double() for t in (Q, K, V))
    S = (q @ k.transpose(-1, -2)) * scale
    if causal:
        S = S.masked_fill(~torch.ones(N, N, dtype=torch.bool, device=DEV).tril(), float("-inf"))
    O_t = torch.softmax(S, dim=-1) @ v
    L_t = torch.logsumexp(S, dim=-1, keepdim=True) * math.log2(math.e)
    assert torch.isfinite(O.float()).all() and torch.isfinite(L.float()).all()
    step = 2.0 ** -3 if dtype == torch.float8_e4m3fn else 2.0 ** -2        # one ulp, relative
    rel = ((O.double() - O_t).abs() / O_t.abs().clamp(min=0.05)).flatten()
    assert rel.median().item() <= step / 2 and rel.kthvalue(int(0.99 * rel.numel())).values.item() <= 3 * step, \
        (rel.median().item(), rel.kthvalue(int(0.99 * rel.numel())).values.item())
    assert ((L.double() - L_t).abs() <= step * L_t.abs() + 1.5 * step).all()


@pytest.mark.parametrize("B,H,N,dtype,causal,scale,layout", a64_cases()[:16], ids=lambda v: str(v).replace("torch.", ""))
def test_random_a64d_problem(B, H, N, dtype, causal, scale, layout):
    """the generated kernel at head size 64 over the same random problems (ragged N or not, scales, storage layouts)"""
    g = torch.Generator().manual_seed(N * 7919 + B * 31 + H + 64)
    mk = {"contiguous": lambda: (torch.randn(B, H, N, 64, generator=g) * 0.7).to(dtype).to(DEV),
          "bnhd": lambda: (torch.randn(B, N, H, 64, generator=g) * 0.7).to(dtype).to(DEV).transpose(1, 2),
          "padded_rows": lambda: (torch.randn(B, H, N, 72, generator=g) * 0.7).to(dtype).to(DEV)[..., :64]}[layout]
    Q, K, V = mk(), mk(), mk()
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal, scale=scale, variant="a64d")
    q, k, v = (t.double() for t in (Q, K, V))
    S = (q @ k.transpose(-1, -2)) * scale
    if causal:
        S = S.masked_fill(~torch.ones(N, N, dtype=torch.bool, device=DEV).tril(), float("-inf"))
    O_t = torch.softmax(S, dim=-1) @ v
    L_t = torch.logsumexp(S, dim=-1, keepdim=True) * math.log2(math.e)
    assert (O.double() - O_t).abs().max().item() <= REL[dtype] * max(1.0, O_t.abs().max().item())
    lu = 2.0 ** (math.floor(math.log2(max(L_t.abs().max().item(), 1e-9))) - (7 if dtype == torch.bfloat16 else 10))
    assert (L.double() - L_t).abs().max().item() <= 1.01 * lu
